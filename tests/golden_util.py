"""Loading of the committed golden fixtures (tests/golden/*.npz, made by tools/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = ["stime", "width", "height", "iter_count", "bounce_count", "ray_count", "light_count", "range", "max_cost_default",
          "debug_nx", "debug_ny", "debug_nz", "debug_scale", "debug_x", "debug_y", "debug_z", "show_objects"]


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def scene_of(path):
    return os.path.basename(path).rsplit("_t", 1)[0]


def oracle_frame(oracle, g):
    f = oracle.OrcFrame()
    for k in FIELDS:
        v = g[k].item()
        setattr(f, k, int(v) if k in ("width", "height", "iter_count", "bounce_count", "ray_count", "light_count", "max_cost_default") else float(v))
    b = g["basis"]
    for i in range(3):
        f.eye[i], f.front[i], f.right[i], f.top[i] = b[0][i], b[1][i], b[2][i], b[3][i]
    for i in range(8):
        f.scene_var[i] = g["scene_var"][i]
    # the fixtures were made at the reference's epsilons (pshader_sdf.hlsl:31-35)
    f.dist_eps, f.grad_eps, f.reflect_eps, f.refract_eps, f.shadow_eps = 0.0001, 0.0001, 0.001, 0.001, 0.0003
    return f
