"""Developer timing: how long does ONE rank's share of a frame take on one GPU?  (What bounds the scaling of the strip
sharding before any byte moves: rank r of N renders every N-th 8-row strip; N = 1 is the whole frame.)

    python tools/strip_time.py [--config 3] [--worlds 1,2,4,8] [--launch auto|per_tile|persistent]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import sdf_playground_amd as sp


def pipelined(a):
    import time

    cfg = bench.CONFIGS[a.config]
    W, H = cfg["width"], cfg["height"]
    fmt = sp.STRIP_RGB16F_A8 if a.wire == "f16" else sp.STRIP_RGB32F_A8
    mode = {"auto": sp.LAUNCH_AUTO, "per_tile": sp.LAUNCH_PER_TILE, "persistent": sp.LAUNCH_PERSISTENT}[a.launch]
    hs, streams = [], []
    for k in range(a.in_flight):
        st = torch.cuda.Stream()
        h = sp.SDFRenderer(0)
        h.initShader(cfg["scene"])
        h.setLimits(**cfg["limits"])
        h.setLaunchMode(mode)
        h.setStream(st.cuda_stream)
        hs.append(h)
        streams.append(st)
    full = None
    for world in [int(x) for x in a.worlds.split(",")]:
        nb = sp.strip_buffer_bytes(W, H, world, fmt)
        bufs = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in hs]
        frames = 48
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(frames if timed else 8):
                h = hs[k % len(hs)]
                cam, stime = bench.make_camera(k % bench.SWEEP, W, H, a.config)
                h.setParameters(stime)
                h.setCamera(cam)
                h.renderStrips(W, H, 0, world, bufs[k % len(hs)], fmt=fmt)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / frames * 1e3
        if world == 1:
            full = dt
        print("config %s world %d, rank 0, %d frames in flight (%s launch): %.3f ms per frame = %.2fx of 1/%d of the full-frame rate (%.3f ms) -> scaling bound %.2fx" % (
            a.config, world, len(hs), a.launch, dt, dt / (full / world), world, full, full / dt), flush=True)
    for h in hs:
        h.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--launch", default="per_tile")
    ap.add_argument("--wire", default="f16")
    ap.add_argument("--in-flight", type=int, default=1, help="handles / streams the frames of ONE rank alternate between (wall-clock rate over 48 frames)")
    a = ap.parse_args()
    if a.in_flight > 1:
        return pipelined(a)
    cfg = bench.CONFIGS[a.config]
    W, H = cfg["width"], cfg["height"]
    r = sp.SDFRenderer(0)
    r.initShader(cfg["scene"])
    r.setLimits(**cfg["limits"])
    r.setLaunchMode({"auto": sp.LAUNCH_AUTO, "per_tile": sp.LAUNCH_PER_TILE, "persistent": sp.LAUNCH_PERSISTENT}[a.launch])
    fmt = sp.STRIP_RGB16F_A8 if a.wire == "f16" else sp.STRIP_RGB32F_A8
    full = None
    for world in [int(x) for x in a.worlds.split(",")]:
        nb = sp.strip_buffer_bytes(W, H, world, fmt)
        buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
        per_rank = []
        for rank in range(world):
            ms = []
            for k in range(6):
                cam, stime = bench.make_camera(k, W, H, a.config)
                r.setParameters(stime)
                r.setCamera(cam)
                r.renderStrips(W, H, rank, world, buf, fmt=fmt)
                ms.append(r.getStats().ms_gpu)
            per_rank.append(float(np.mean(ms[1:])))
        worst = max(per_rank)
        if world == 1:
            full = worst
        print("config %s world %d (%s launch): rank times %s ms; slowest %.3f ms = %.2fx of 1/%d of the full frame (%.3f ms) -> scaling bound %.2fx" % (
            a.config, world, a.launch, " ".join("%.3f" % m for m in per_rank), worst, worst / (full / world), world, full, full / worst), flush=True)
    r.close()


main()
