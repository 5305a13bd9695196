"""Operation census of the bench workload with the counting build of the oracle
(oracle/liboracle_census.so): algorithmic flops per ray = the roofline numerator.
Writes profiles/census_r01.json.  Sample: all 16 sweep frames, every 16th pixel in x and y."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from oracle import pyoracle as po

po.build(census=True, ref=False)
STEP = 16
tot = np.zeros(4, np.float64)
flops = transc = 0
for k in range(bench.SWEEP):
    eye, direction, stime = bench.sweep_camera(k)
    basis = po.camera_direction(eye, direction, np.float32(bench.sp.to_radian(60.0)), np.float32(bench.WIDTH) / np.float32(bench.HEIGHT))
    f = po.default_frame(bench.SCENE, bench.WIDTH, bench.HEIGHT, basis=basis, stime=stime)
    f.iter_count = bench.ITER_COUNT
    _, t, fl, tr = po.census(bench.SCENE, f, step=(STEP, STEP))
    tot += t.astype(np.float64)
    flops += fl
    transc += tr
px, rays, evals, hits = tot
out = {
    "labyrinth_4k_iter256": {
        "sample": "16 sweep frames, every %dth pixel in x and y (%d pixels)" % (STEP, px),
        "counting_rule": "+ - * / sqrt rsqrt floor round min max compare select = 1, fma = 2, transcendental (sin cos atan2 exp2 log2) = 1, abs/neg = 0",
        "flops_per_ray": flops / rays,
        "flops_per_pixel": flops / px,
        "transcendentals_per_ray": transc / rays,
        "rays_per_pixel": rays / px,
        "march_evals_per_ray": evals / rays,
        "hits_per_ray": hits / rays,
        "flops_per_scene_eval_all_in": flops / (evals + 4 * hits),
    }
}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "census_r01.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
