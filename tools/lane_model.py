"""Developer analysis (CPU only): how well do the lanes of a wave fill with this scene's rays?

Runs the product's per-pixel pipeline on the CPU (tests/hostsim) over a sample of 8x8 tiles of a bench
frame, recording for every ray of every pixel its march iterations, kind and outcome, and evaluates the
lane utilisation of the march loops under different schedules of the same rays:
  pixel    the shipped schedule: bounce b of a tile marches every pixel's b-th ray together (a wave
           iterates max-over-lanes times, lanes whose pixel has no b-th ray idle)
  refill   pixels whose rays ran out are replaced by the next pixel of the wave's tile stream at bounce
           boundaries (upper bound: perfect packing of ray counts, same per-bounce max rule)
  sorted   as pixel, but the rays of a bounce are marched longest-first in ideal 64-wide groups over
           several tiles (what a perfect compaction across tiles could reach)
It prints march work (lane-iterations) against issued wave-iterations x 64.

    python tools/lane_model.py --config 3 [--frame 5] [--step 4]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3")
    ap.add_argument("--frame", type=int, default=5)
    ap.add_argument("--step", type=int, default=4, help="every step-th tile in x and y")
    ap.add_argument("--scene", default=None, help="instead of a configuration: this scene as tools/scene_table.py renders it (default camera, 3840x2160, 256 steps)")
    a = ap.parse_args()
    if a.scene:
        cfg = {"scene": a.scene}
        W, H = 3840, 2160
        of = po.default_frame(a.scene, W, H, stime=a.frame / 60.0)
        of.iter_count = 256
        f = hostsim.frame_from_oracle(of)
        a.config = a.scene
    else:
        cfg = bench.CONFIGS[a.config]
        W, H = cfg["width"], cfg["height"]
        f = hostsim.frame_from_oracle(bench.oracle_frame(po, a.frame, W, H, a.config))
    L = hostsim.lib()
    L.hostsim_trace_tiles.restype = ctypes.c_longlong
    L.hostsim_trace_tiles.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
    ntiles = ((W + 7) // 8 + a.step - 1) // a.step * (((H + 7) // 8 + a.step - 1) // a.step)
    rec = np.zeros((ntiles, 64, 16), np.uint32)
    n = L.hostsim_trace_tiles(cfg["scene"].encode(), ctypes.byref(f), a.step, rec.ctypes.data_as(ctypes.c_void_p), ntiles, os.cpu_count() or 1)
    assert n == ntiles, n
    evals = (rec & 0xffff).astype(np.int64)          # [tile, lane, bounce]
    used = rec != 0
    shadow = ((rec >> 20) & 1).astype(bool)
    rays = used.sum()
    print("config %s frame %d: %d tiles sampled, %.2f rays/pixel, %.1f march iterations/ray" % (a.config, a.frame, ntiles, rays / (ntiles * 64.0), evals.sum() / rays))
    work = evals.sum()
    # pixel schedule
    per_bounce_max = evals.max(axis=1)               # [tile, bounce]
    issued = per_bounce_max.sum() * 64
    print("  pixel   : lane-iterations %d, issued %d, utilisation %.3f" % (work, issued, work / issued))
    # where the idle lanes are: within-bounce length spread vs lanes without a ray
    has = used.sum(axis=1)                           # [tile, bounce] lanes with a ray
    issued_only_present = (per_bounce_max * has).sum()
    print("            of the idle lane-iterations: %.1f %% lanes without a ray in that bounce, %.1f %% shorter rays waiting for the longest" % (
        100.0 * (issued - issued_only_present) / (issued - work), 100.0 * (issued_only_present - work) / (issued - work)))
    for b in range(6):
        m = used[:, :, b]
        if m.sum() == 0:
            break
        print("            bounce %d: %5.1f %% of lanes have a ray (%4.1f %% of them shadow rays), mean %5.1f iterations, mean of tile maxima %5.1f" % (
            b, 100.0 * m.mean(), 100.0 * shadow[:, :, b][m].mean(), evals[:, :, b][m].mean(), per_bounce_max[:, b][per_bounce_max[:, b] > 0].mean()))
    # refill at bounce boundaries: every bounce's rays of all tiles packed into full waves in tile order (keeps neighbours together)
    issued_refill = 0
    for b in range(16):
        e = evals[:, :, b][used[:, :, b]]            # tile-major order
        if e.size == 0:
            continue
        pad = (-e.size) % 64
        e = np.concatenate([e, np.zeros(pad, np.int64)]).reshape(-1, 64)
        issued_refill += e.max(axis=1).sum() * 64
    print("  refill  : issued %d, utilisation %.3f  (rays of a bounce packed 64 at a time in tile order)" % (issued_refill, work / issued_refill))
    issued_sorted = 0
    for b in range(16):
        e = np.sort(evals[:, :, b][used[:, :, b]])[::-1]
        if e.size == 0:
            continue
        pad = (-e.size) % 64
        e = np.concatenate([e, np.zeros(pad, np.int64)]).reshape(-1, 64)
        issued_sorted += e.max(axis=1).sum() * 64
    print("  sorted  : issued %d, utilisation %.3f  (the same, longest rays first: an upper bound)" % (issued_sorted, work / issued_sorted))
    # all rays regardless of bounce, persistent refill of single lanes (ideal dynamic): utilisation -> 1 minus tail; report mean/max spread instead
    print("  ideal per-lane refill (a lane takes its pixel's next ray the moment one ends, no shading stalls): 1.000 by construction")


main()
