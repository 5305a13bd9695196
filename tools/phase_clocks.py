"""Developer tool: where does a wave of the pixel kernel spend its clocks?

Builds libsdfr with -DSDFR_PHASE_CLOCKS into gpurun_out/libsdfr_clocks.so (the render totals
then carry wave clocks: whole pixel loop / marching / background / normals + shading, from the lane of each
wave that stayed longest) and prints the split for a few frames.  Not a product build.

  python tools/phase_clocks.py --build          (here: cross-compile)
  python tools/phase_clocks.py --scene labyrinth (on the GPU box)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libsdfr_clocks.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--scene", default="labyrinth")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--iters", type=int, default=256)
    ap.add_argument("--config", default=None, help="a BASELINE configuration of bench.py (2, 3, 4, 5, 5g): its scene, size, limits and camera sweep")
    a = ap.parse_args()
    if a.build:
        from sdf_playground_amd import buildlib
        print(buildlib.build(force=True, extra=["-DSDFR_PHASE_CLOCKS"], out=LIB))
        return
    os.environ["SDFR_LIBRARY"] = LIB
    import torch
    import sdf_playground_amd as sp
    from quickbench import camera_for

    cfg = None
    if a.config:
        import bench
        cfg = bench.CONFIGS[a.config]
        a.scene, a.width, a.height = cfg["scene"], cfg["width"], cfg["height"]
    r = sp.SDFRenderer(0)
    r.initShader(a.scene)
    r.setSchedule(1)
    r.setLimits(iter_count=a.iters)
    if cfg:
        r.setLimits(**cfg["limits"])
    out = torch.empty((a.height, a.width, 4), dtype=torch.float32, device="cuda")
    for k in range(a.frames):
        if cfg:
            cam, stime = bench.make_camera(k, config=a.config)
            r.setParameters(stime)
            r.render(cam, a.width, a.height, out=out)
        else:
            r.setParameters(k / 60.0)
            r.render(camera_for(a.scene, k, a.width, a.height), a.width, a.height, out=out)
        s = r.getStats()
        tot = max(1, s.pixels)
        other = s.pixels - s.rays - s.march_evals - s.hits
        print("%s frame %d: %.3f ms; wave clocks: march %.1f%%, background %.1f%%, normals+shading %.1f%%, queue/other %.1f%%" % (
            a.scene, k, s.ms_gpu, 100.0 * s.rays / tot, 100.0 * s.march_evals / tot, 100.0 * s.hits / tot, 100.0 * other / tot), flush=True)
    r.close()


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    main()
