"""Developer analysis (CPU only): what would lane re-packing buy?

Takes the per-ray traces of tools/lane_model.py (every ray of every pixel of a sample of 8x8 tiles: march
iterations, outcome, kind) and prices two schedules of the SAME rays in VALU instructions issued per wave:
  tile     the shipped pixel kernel: a wave renders one tile at a time, bounce b of all its pixels together
  repack   a lane is a pixel stream: when its ray ends it waits; once `T` lanes wait (or nobody marches) the
           waiting lanes are shaded together, pop their next ray, or take the next pixel of the wave's tile stream
Costs (instructions per wave-level execution): E per march step, S_hit = normal + material + lights, S_sh = a
shadow ray's hit (material only), S_bg = background of an escaped ray, A = queue pop + ray set-up, P = new pixel.

    python tools/repack_model.py --config 5 --E 212 --hit 2100 --sh 400 --bg 1300 [--T 16,24,32,48]
    python tools/repack_model.py --scene tree --E 1884 --hit 8000 --sh 2000 --bg 1300
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import bench
import hostsim
from oracle import pyoracle as po


def trace(a):
    if a.scene:
        scene, W, H = a.scene, 3840, 2160
        of = po.default_frame(scene, W, H, stime=a.frame / 60.0)
        of.iter_count = 256
    else:
        cfg = bench.CONFIGS[a.config]
        scene, W, H = cfg["scene"], cfg["width"], cfg["height"]
        of = bench.oracle_frame(po, a.frame, W, H, a.config)
    f = hostsim.frame_from_oracle(of)
    L = hostsim.lib()
    L.hostsim_trace_tiles.restype = ctypes.c_longlong
    L.hostsim_trace_tiles.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
    ntiles = ((W + 7) // 8 + a.step - 1) // a.step * (((H + 7) // 8 + a.step - 1) // a.step)
    rec = np.zeros((ntiles, 64, 16), np.uint32)
    n = L.hostsim_trace_tiles(scene.encode(), ctypes.byref(f), a.step, rec.ctypes.data_as(ctypes.c_void_p), ntiles, os.cpu_count() or 1)
    assert n == ntiles, n
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--frame", type=int, default=5)
    ap.add_argument("--step", type=int, default=6)
    ap.add_argument("--E", type=float, default=324)
    ap.add_argument("--hit", type=float, default=3500)
    ap.add_argument("--sh", type=float, default=600)
    ap.add_argument("--bg", type=float, default=1300)
    ap.add_argument("--A", type=float, default=100)
    ap.add_argument("--P", type=float, default=200)
    ap.add_argument("--T", default="8,16,24,32,48,64")
    ap.add_argument("--waves", type=int, default=64, help="the sampled tiles are dealt to this many model waves in turn")
    a = ap.parse_args()
    rec = trace(a)
    evals = (rec & 0xffff).astype(np.int64)
    used = rec != 0
    status_hit = ((rec >> 16) & 0xf) == 1   # MARCH_HIT
    shadow = ((rec >> 20) & 1).astype(bool)
    ntiles = rec.shape[0]
    work = float(evals.sum())

    # the shipped schedule
    cost = 0.0
    steps = 0
    for b in range(16):
        u = used[:, :, b]
        if not u.any():
            break
        mx = evals[:, :, b].max(axis=1)
        steps += mx.sum()
        cost += (mx * a.E).sum()
        cost += ((u & status_hit[:, :, b] & ~shadow[:, :, b]).any(axis=1) * a.hit).sum()
        cost += ((u & status_hit[:, :, b] & shadow[:, :, b]).any(axis=1) * a.sh).sum()
        cost += ((u & ~status_hit[:, :, b] & ~shadow[:, :, b]).any(axis=1) * a.bg).sum()
        cost += (u.any(axis=1) * a.A).sum() + (a.P * ntiles if b == 0 else 0)
    print("tile    : march utilisation %.3f, %.0f instructions per pixel, march share %.2f" % (work / (steps * 64.0), cost / (ntiles * 64.0), steps * a.E / cost))
    base = cost

    nrays = used.sum(axis=2)  # [tile, lane]
    for T in [int(x) for x in a.T.split(",")]:
        total_cost, total_steps, batches = 0.0, 0, 0
        for w in range(a.waves):
            mine = list(range(w, ntiles, a.waves))
            # pixel stream of the wave: tiles in order, lanes in order
            stream = [(t, l) for t in mine for l in range(64)]
            pos = 0
            lane_pix = [None] * 64   # (tile, lane)
            lane_ray = [0] * 64      # index of the current ray
            lane_left = [0] * 64     # march steps left; 0 and lane_pix set = waiting for shade
            state = ["need_pixel"] * 64
            while True:
                # A: waiting lanes were shaded: next ray or next pixel
                took_pixel = False
                for l in range(64):
                    if state[l] == "need_ray":
                        t, pl = lane_pix[l]
                        lane_ray[l] += 1
                        if lane_ray[l] < nrays[t, pl]:
                            lane_left[l] = int(evals[t, pl, lane_ray[l]])
                            state[l] = "march"
                        else:
                            state[l] = "need_pixel"
                    if state[l] == "need_pixel":
                        if pos < len(stream):
                            lane_pix[l] = stream[pos]
                            pos += 1
                            lane_ray[l] = 0
                            lane_left[l] = int(evals[lane_pix[l][0], lane_pix[l][1], 0])
                            state[l] = "march"
                            took_pixel = True
                        else:
                            state[l] = "done"
                if all(s == "done" for s in state):
                    break
                total_cost += a.A + (a.P if took_pixel else 0)
                # B: march until T lanes wait or nobody marches
                while True:
                    marching = [l for l in range(64) if state[l] == "march"]
                    waiting = sum(1 for s in state if s == "wait")
                    if not marching or waiting >= T:
                        break
                    # advance to the next event: the smallest remaining count among the marching lanes
                    k = min(lane_left[l] for l in marching)
                    total_steps += k
                    total_cost += k * a.E
                    for l in marching:
                        lane_left[l] -= k
                        if lane_left[l] == 0:
                            state[l] = "wait"
                # C: shade the waiting lanes
                kinds = set()
                for l in range(64):
                    if state[l] == "wait":
                        t, pl = lane_pix[l]
                        r = lane_ray[l]
                        if status_hit[t, pl, r]:
                            kinds.add("sh" if shadow[t, pl, r] else "hit")
                        elif not shadow[t, pl, r]:
                            kinds.add("bg")
                        state[l] = "need_ray"
                total_cost += (a.hit if "hit" in kinds else 0) + (a.sh if "sh" in kinds else 0) + (a.bg if "bg" in kinds else 0)
                batches += 1
        print("repack T=%2d: march utilisation %.3f, %.0f instructions per pixel (%.2fx of tile), %.1f shade batches per 64 pixels" % (
            T, work / (total_steps * 64.0), total_cost / (ntiles * 64.0), base / total_cost, batches / float(ntiles)))


main()
