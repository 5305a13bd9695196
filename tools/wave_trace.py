"""Developer tool: when and where do the waves of the pixel kernel run?

Builds libsdfr with -DSDFR_WAVE_TRACE into tools/libsdfr_trace.so (every wave's per-block record then
carries its start / end time in 10-ns ticks, its HW_ID / XCC_ID and its march evaluations instead of
the render counters) and prints, for a few frames of a bench configuration: the occupancy curve
(resident waves over time), the share of the frame spent below half occupancy, the distribution of
wave lifetimes and how evenly the XCDs finish.  Not a product build.

  python tools/wave_trace.py --build            (here: cross-compile)
  python tools/wave_trace.py --config 3         (on the GPU box)
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libsdfr_trace.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--config", default="3")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "wave_trace"))
    a = ap.parse_args()
    if a.build:
        from sdf_playground_amd import buildlib
        print(buildlib.build(force=True, extra=["-DSDFR_WAVE_TRACE"], out=LIB))
        return
    os.environ["SDFR_LIBRARY"] = LIB
    import numpy as np
    import torch
    import bench
    import sdf_playground_amd as sp

    cfg = bench.CONFIGS[a.config]
    W, H = cfg["width"], cfg["height"]
    L = sp.load_library()
    L.sdfr_debug_read_partials.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    r = sp.SDFRenderer(0)
    r.initShader(cfg["scene"])
    r.setLimits(**cfg["limits"])
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    nblocks = ((W + 7) // 8) * ((H + 7) // 8)
    os.makedirs(a.out, exist_ok=True)
    for k in range(a.frames + 1):
        cam, stime = bench.make_camera(k, W, H, a.config)
        r.setParameters(stime)
        r.render(cam, W, H, out=out)
        ms = r.getStats().ms_gpu
        rec = np.zeros((nblocks, 4), np.uint64)
        assert L.sdfr_debug_read_partials(r._h, rec.ctypes.data_as(ctypes.c_void_p), nblocks) == 0
        if k == 0:
            continue  # warm-up (and the read above has cleared the records: waves that do not exist read as zero)
        rec = rec[(rec[:, 0] > 0) & (rec[:, 1] >= rec[:, 0])]
        nwaves = len(rec)
        t0 = rec[:, 0].astype(np.int64)
        t1 = rec[:, 1].astype(np.int64)
        if nwaves == 0 or (t1.max() - t0.min()) > 10_000_000:  # > 0.1 s: not a trace of one frame
            print("frame %d: no usable trace (%d records)" % (k, nwaves))
            continue
        base = t0.min()
        t0, t1 = (t0 - base) * 0.01, (t1 - base) * 0.01  # microseconds
        T = t1.max()
        life = t1 - t0
        hw = rec[:, 2]
        xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
        # occupancy curve at 1-us resolution: +1 at start, -1 at end
        n = int(T) + 2
        d = np.zeros(n + 1, np.int64)
        np.add.at(d, t0.astype(np.int64), 1)
        np.add.at(d, np.minimum(t1.astype(np.int64) + 1, n), -1)
        occ = np.cumsum(d)[:n]
        peak = occ.max()
        tiles_done = (rec[:, 3] & np.uint64(0xff)).astype(np.int64)
        first_claim_us = ((rec[:, 3] >> np.uint64(8)) & np.uint64(0xffffff)).astype(np.int64) * 0.01
        print("config %s frame %d: event %.3f ms; first start -> last end %.1f us; %d waves, %d tiles (per wave min %d mean %.1f max %d); peak resident %d; mean resident %.0f (%.2f of peak)" % (
            a.config, k, ms, T, nwaves, tiles_done.sum(), tiles_done.min(), tiles_done.mean(), tiles_done.max(), peak, occ.mean(), occ.mean() / peak))
        for frac in (0.9, 0.5, 0.25):
            below = (occ < frac * peak).sum()
            print("   time below %.0f %% of peak occupancy: %6.1f us (%.1f %% of the frame)" % (100 * frac, below, 100.0 * below / n))
        # ramp: when does occupancy first reach 90 % of peak, when does it last hold it
        hi = np.nonzero(occ >= 0.9 * peak)[0]
        print("   ramp-up to 90 %%: %.1f us; last time at 90 %%: %.1f us (tail %.1f us)" % (hi[0], hi[-1], T - hi[-1]))
        q = np.percentile(life, [5, 25, 50, 75, 95, 99, 100])
        print("   wave lifetime us: p5 %.1f p25 %.1f p50 %.1f p75 %.1f p95 %.1f p99 %.1f max %.1f; sum of lifetimes / (T x peak) = %.3f" % (*q, life.sum() / (T * peak)))
        tile_us = (rec[:, 3] >> np.uint64(32)).astype(np.int64) * 0.01
        print("   time between taking a tile and having rendered it, summed / wave lifetimes summed: %.4f (the rest: waiting for a tile, start-up, wind-down); per tile not rendering: %.2f us" % (
            tile_us.sum() / life.sum(), (life.sum() - tile_us.sum()) / max(1, tiles_done.sum())))
        had = tiles_done >= 0
        print("   from a wave's first instruction to having its first tile, us: p5 %.1f p50 %.1f p95 %.1f mean %.1f (x %d waves = %.1f %% of the lifetimes)" % (
            *np.percentile(first_claim_us[had], [5, 50, 95]), first_claim_us[had].mean(), nwaves, 100.0 * first_claim_us.sum() / life.sum()))
        ends = [t1[xcc == x].max() for x in range(8) if (xcc == x).any()]
        cnt = [int((xcc == x).sum()) for x in range(8)]
        print("   per-XCD last wave end (us): %s; waves per XCD: %s" % (" ".join("%.0f" % e for e in ends), cnt))
        # launch order vs start time: how far ahead of the finishing front does the dispatcher run
        print("   wave end time quantiles (us): %s" % " ".join("%.0f" % q for q in np.percentile(t1, [0, 1, 5, 25, 50, 75, 95, 99, 100])))
        late = np.argsort(t1)[-8:]
        print("   the 8 last waves: " + "; ".join("[%.0f..%.0f] tiles %d rendering %.0f us" % (t0[b], t1[b], int(tiles_done[b]), 0.01 * int(rec[b, 3] >> np.uint64(32))) for b in late))
        np.savez_compressed(os.path.join(a.out, "cfg%s_frame%d.npz" % (a.config, k)), rec=rec, ms=ms)
    r.close()


if __name__ == "__main__":
    main()
