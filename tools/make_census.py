"""Operation census of the bench workloads with the counting build of the oracle
(oracle/liboracle_census.so): algorithmic flops per ray = the roofline numerator of bench.py.
Writes profiles/census_<round>.json, one entry per BASELINE configuration (bench.CONFIGS).
Sample: all 16 sweep frames, every STEPth pixel in x and y.

    python tools/make_census.py [config ...]      (default: every configuration)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from oracle import pyoracle as po

po.build(census=True, ref=False)
STEP = {"2": 8, "1": 1}  # 1920x1080: a denser sample for a similar pixel count; 256x256: every pixel
configs = sys.argv[1:] or sorted(bench.CONFIGS)
try:
    with open(bench.CENSUS_FILE) as fh:
        out = json.load(fh)
except Exception:
    out = {}
for c in configs:
    cfg = bench.CONFIGS[c]
    step = STEP.get(c, 16)
    tot = np.zeros(4, np.float64)
    flops = transc = 0
    for k in range(bench.SWEEP):
        f = bench.oracle_frame(po, k, cfg["width"], cfg["height"], c)
        _, t, fl, tr = po.census(cfg["scene"], f, step=(step, step))
        tot += t.astype(np.float64)
        flops += fl
        transc += tr
    px, rays, evals, hits = tot
    out[cfg["key"]] = {
        "config": c,
        "workload": cfg["workload"] % (cfg["width"], cfg["height"]),
        "sample": "16 sweep frames, every %dth pixel in x and y (%d pixels)" % (step, px),
        "counting_rule": "+ - * / sqrt rsqrt floor round min max compare select = 1, fma = 2, transcendental (sin cos atan2 exp2 log2) = 1, abs/neg = 0",
        "flops_per_ray": flops / rays,
        "flops_per_pixel": flops / px,
        "transcendentals_per_ray": transc / rays,
        "rays_per_pixel": rays / px,
        "march_evals_per_ray": evals / rays,
        "hits_per_ray": hits / rays,
        "flops_per_scene_eval_all_in": flops / (evals + 4 * hits),
    }
    print(c, json.dumps(out[cfg["key"]], indent=1), flush=True)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(bench.CENSUS_FILE, "w") as fh:
    json.dump(out, fh, indent=1)
