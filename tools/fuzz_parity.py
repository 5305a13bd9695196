"""Randomised parity hunt (GPU box): random cameras, times, variable values and limits for every
built-in scene, HIP kernels against the CPU oracle, bit for bit (pixels and per-pixel counters).
The fixed-camera tests cannot find a culling bound or a fast-math domain that only fails from
some other viewpoint; this can.  Prints every mismatch; exit status 1 if there was one.

    python tools/fuzz_parity.py --cases 40 --seed 1"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import sdf_playground_amd as sp
from oracle import pyoracle as po


def run(cases, seed, size, scenes=None, verbose=False, shortcut_heavy=False):
    rng = np.random.default_rng(seed)
    W, H = size
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    asp = np.float32(W) / np.float32(H)
    r = sp.SDFRenderer(0)
    bad = []
    n = 0
    for scene in (scenes or sp.scene_names()):
        r.initShader(scene)
        table = po.var_table(scene)
        for c in range(cases):
            # camera: somewhere in a box around the origin, looking roughly at the scene's middle
            eye = (float(rng.uniform(-9, 9)), float(rng.uniform(0.2, 8)), float(rng.uniform(-9, 9)))
            tgt = (float(rng.uniform(-2, 2)), float(rng.uniform(0, 3)), float(rng.uniform(-2, 2)))
            if c % 5 == 4:  # sometimes from far away / from below the canopy / grazing the floor
                eye = (float(rng.uniform(-40, 40)), float(rng.choice([0.05, 0.5, 25.0])), float(rng.uniform(-40, 40)))
            if c % 7 == 6:  # under the floor, exactly on it, a hair above it (the fast plane divides the height by 1e-20)
                eye = (eye[0], float(rng.choice([-1.0, -0.05, 0.0, 1e-22, 1e-6])), eye[2])
            stime = float(np.float32(rng.uniform(0, 30)))
            basis = po.camera_lookat(eye, tgt, fovy, asp)
            f = po.default_frame(scene, W, H, basis=basis, stime=stime)
            limits = dict(iter_count=int(rng.choice([100, 100, 256, 37])), max_cost_default=int(rng.choice([7, 7, 9, 4])),
                          ray_count=int(rng.choice([8, 8, 3])), bounce_count=int(rng.choice([16, 16, 5])),
                          light_count=8, range=100.0, extension_lights=int(rng.choice([0, 0, 0, 7])),
                          extension_marble_reflection=float(rng.choice([0.0, 0.0, 0.0, 0.25])))
            # the driver's epsilons: the reference's, or (one case in four) all five somewhere in their accepted ranges
            if rng.random() < 0.25:
                limits.update(dist_eps=float(np.float32(rng.choice([1e-5, 3e-4, 1e-3]))), grad_eps=float(np.float32(rng.choice([1e-5, 1e-3, 2e-2]))),
                              reflect_eps=float(np.float32(rng.choice([0.0, 1e-4, 1e-2]))), refract_eps=float(np.float32(rng.choice([0.0, 1e-4, 1e-2]))),
                              shadow_eps=float(np.float32(rng.choice([0.0, 1e-4, 5e-3]))))
            else:
                limits.update(dist_eps=0.0001, grad_eps=0.0001, reflect_eps=0.001, refract_eps=0.001, shadow_eps=0.0003)
            for k, v in limits.items():
                setattr(f, k, v)
            r.setLimits(**limits)
            r.setParameters(stime)
            cam = sp.Camera()
            cam.SetEye(eye)
            cam.SetLookat(tgt)
            cam.SetFOVY(float(fovy))
            cam.SetAspect(float(asp))
            r.resetVariables()
            values = {}
            for name, mn, mx, start, _st, _v, slot in table:
                if slot >= 0 and rng.random() < 0.7:
                    v = float(np.float32(rng.uniform(mn, mx)))
                    values[name] = v
                    f.scene_var[slot] = v
                    r.setValue(name, v)
            schedule = int(rng.integers(0, 2))
            if shortcut_heavy:  # the pixel schedule with step shortcuts on, eight lights in half of the cases: what the escape rules and the delivered shadow rays see
                schedule = 1
                f.extension_lights = limits["extension_lights"] = int(rng.choice([0, 7]))
                if rng.random() < 0.5:  # any ray budget and queue length: the delivered shadow rays have to respect both
                    f.bounce_count = limits["bounce_count"] = int(rng.integers(1, 17))
                    f.ray_count = limits["ray_count"] = int(rng.integers(1, 9))
                r.setLimits(**limits)
            r.setSchedule(schedule)
            r.setLaunchMode(int(rng.choice([sp.LAUNCH_AUTO, sp.LAUNCH_PER_TILE, sp.LAUNCH_PERSISTENT])))  # pixel schedule: how the tiles reach the waves
            shortcuts = (bool(rng.integers(0, 2)) or shortcut_heavy) and schedule == 1  # step shortcuts: same pixels, rays and hits; fewer steps counted
            r.setStepShortcuts(shortcuts)
            img, st = r.render(cam, W, H, pixel_stats=True)
            ref, rst, _ = po.render(scene, f, stats=True)
            n += 1
            stats_same = np.array_equal(st, rst) if not shortcuts else (np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]) and bool((st[..., 1] <= rst[..., 1]).all()))
            same = np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and stats_same
            if not same:
                # NaN payloads may differ in sign/payload bits: compare values with NaN == NaN as well
                same = np.array_equal(img, ref, equal_nan=True) and stats_same
            if not same:
                diff = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
                bad.append((scene, c, eye, tgt, stime, limits, values, schedule, diff))
                print("MISMATCH", bad[-1], flush=True)
            elif verbose:
                print("ok", scene, c, flush=True)
        print("%-20s %d cases done, %d mismatches so far" % (scene, cases, len(bad)), flush=True)
    r.close()
    return n, bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--size", default="64x48")
    ap.add_argument("--scenes", default="")
    ap.add_argument("--shortcut-heavy", action="store_true", help="every case on the pixel schedule with step shortcuts, half of them with eight lights")
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    n, bad = run(a.cases, a.seed, (w, h), [s for s in a.scenes.split(",") if s] or None, shortcut_heavy=a.shortcut_heavy)
    print("%d cases, %d mismatches" % (n, len(bad)))
    sys.exit(1 if bad else 0)
