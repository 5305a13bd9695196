"""Developer timing helper (not the contract bench): renders a scene a few times per schedule
and prints ms/frame, Mrays/s and the march/shade split."""
import argparse
import math
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdf_playground_amd as sp


def camera_for(scene, k, w, h):
    cam = sp.Camera()
    cam.SetAspect(w / h)
    th = 2 * math.pi * (k + 0.37) / 16
    if scene == "cube_sea":
        cam.SetEye((3 * math.cos(th), 4.5, 3 * math.sin(th)))
        cam.SetDirection((math.cos(th + 0.6), -0.45, math.sin(th + 0.6)))
    elif scene == "labyrinth":
        cam.SetEye((1.5 * math.cos(th), 5.0, 1.5 * math.sin(th)))
        cam.SetDirection((math.cos(th), -0.35, math.sin(th)))
    elif scene == "fractal":
        cam.SetEye((2.2 * math.cos(th), 1.6, 2.2 * math.sin(th)))
        cam.SetLookat((0, 1, 0))
    elif scene == "lense":
        ph = -0.5 + (k + 0.37) / 16
        cam.SetEye((7 * math.sin(ph), 0.5, 7 * math.cos(ph)))
        cam.SetLookat((0, 0, 0))
    elif scene == "gems":
        cam.SetEye((2.5 * math.cos(th), 2, 2.5 * math.sin(th)))
        cam.SetLookat((0, 1, 0))
    elif scene == "light_shadows":
        cam.SetEye((0, 5, -9))
        cam.SetLookat((0, 1, 0))
    return cam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="labyrinth")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--iters", type=int, default=256)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--schedules", default="0,1")
    ap.add_argument("--profile", type=int, default=1)
    a = ap.parse_args()
    r = sp.SDFRenderer(0)
    r.initShader(a.scene)
    r.setLimits(iter_count=a.iters)
    out = torch.empty((a.height, a.width, 4), dtype=torch.float32, device="cuda")
    for sched in [int(x) for x in a.schedules.split(",")]:
        r.setSchedule(sched)
        r.setProfiling(bool(a.profile))
        for k in range(a.frames):
            r.setParameters(k / 60.0)
            r.render(camera_for(a.scene, k, a.width, a.height), a.width, a.height, out=out)
            s = r.getStats()
            print("%s sched=%d frame=%d: %.3f ms, %.1f Mrays/s, rays/px %.2f, evals/ray %.1f, march %.3f ms shade %.3f ms" % (
                a.scene, sched, k, s.ms_gpu, s.rays / s.ms_gpu / 1e3, s.rays / max(1, s.pixels), s.march_evals / max(1, s.rays), s.ms_march, s.ms_shade), flush=True)
    r.close()


if __name__ == "__main__":
    main()
