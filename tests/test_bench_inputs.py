"""CPU tier: the synthetic inputs of bench.py (SURVEY.md 8d cfg 3) are deterministic and
match their definition."""
import math

import numpy as np


def test_pcg_hash_matches_oracle(oracle):
    import bench

    for x in [0, 1, 0x5DF00003, 0x5DF00003 + 15, 0xFFFFFFFF]:
        assert bench.pcg_hash(x) == int(oracle.kat_u32("hash", x).view(np.uint32)[0])


def test_sweep_camera_definition():
    import bench

    for k in range(bench.SWEEP):
        eye, direction, stime = bench.sweep_camera(k)
        u = bench.pcg_hash(bench.SEED + k) / 4294967295.0
        t = 2 * math.pi * (k + u) / 16
        assert eye == (1.5 * math.cos(t), 5.0, 1.5 * math.sin(t))
        assert direction == (math.cos(t), -0.35, math.sin(t))
        assert stime == k / 60.0
        assert abs(eye[0]) < 2 and abs(eye[2]) < 2  # inside the free courtyard, above the walls
    assert bench.sweep_camera(3) == bench.sweep_camera(3)


def test_census_file_matches_workload():
    import json
    import os

    import bench

    with open(bench.CENSUS_FILE) as fh:
        census = json.load(fh)
    c = census["labyrinth_4k_iter256"]
    assert 5e3 < c["flops_per_ray"] < 5e4 and 1.0 < c["rays_per_pixel"] < 3.0
    # one entry per BASELINE configuration, each naming its workload
    for k, cfg in bench.CONFIGS.items():
        e = census[cfg["key"]]
        assert e["config"] == k and e["workload"] == cfg["workload"] % (cfg["width"], cfg["height"]) and e["flops_per_ray"] > 500
        assert bench.load_census(k) == e
