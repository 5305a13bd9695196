"""Generates the committed golden fixtures tests/golden/*.npz with the CPU oracle.

The reference's hot path (HLSL on D3D11) cannot run here and holds no image fixtures
(SURVEY.md 8c), so these vectors are outputs of the oracle -- pinned by the reference's own
known-answer tests and compiled Camera/Math3D -- frozen at the moment the oracle and the
independently written HIP pipeline agreed bit for bit.  They guard both against drift.
Fixture = data only: inputs (camera basis, time, limits, variables) and expected outputs
(fp32 RGBA frame, per-pixel {rays, march evaluations, hits})."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import pyoracle as po

W = H = 64
TH = 0.3
FOVY = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
CAMS = {
    "fast_sphere": ("lookat", (0, 2, -3), (0, 1, 0)),
    "cube_sea": ("dir", (3 * math.cos(TH), 4.5, 3 * math.sin(TH)), (math.cos(TH + 0.6), -0.45, math.sin(TH + 0.6))),
    "labyrinth": ("dir", (1.5 * math.cos(TH), 5.0, 1.5 * math.sin(TH)), (math.cos(TH), -0.35, math.sin(TH))),
    "fractal": ("lookat", (2.2 * math.cos(TH), 1.6, 2.2 * math.sin(TH)), (0, 1, 0)),
    "lense": ("lookat", (7 * math.sin(-0.3), 0.5, 7 * math.cos(-0.3)), (0, 0, 0)),
    "gems": ("lookat", (2.5 * math.cos(TH), 2, 2.5 * math.sin(TH)), (0, 1, 0)),
    "light_shadows": ("lookat", (0, 5, -9), (0, 1, 0)),
    "cube": ("lookat", (2.5, 2.5, -3), (0, 1, 0)),
    "gyroid": ("lookat", (1.8, 1.6, -2.2), (0, 0, 0)),
    "basic_transparency": ("lookat", (2.0, 2.5, -4), (0, 2, 0)),
    "basic_clouds": ("lookat", (0, 2, -8), (0, 4, 0)),
    "coordinate_material": ("lookat", (3, 4, -5), (0, 2, 0)),
    "distortion": ("lookat", (0.8, 1.8, -2.5), (0, 1.5, 0)),
    "table": ("lookat", (2, 2, -3), (0, 1, 0)),
    "sierpinski": ("lookat", (1.2, 1.6, -2.2), (0, 1.2, 0)),
    "neon": ("lookat", (-2, 2.5, -3.5), (0, 2, 1)),
    "fractal2": ("lookat", (1.8, 1.8, -2.0), (0, 1, 0)),
    "shell": ("lookat", (-2.2, 1.8, -2.0), (0, 1, 0)),
    "spiral": ("lookat", (0, 4, -12), (0, 3, 0)),
    "terrain": ("lookat", (6, 5, -8), (0, 0, 0)),
    "tiling": ("lookat", (0, 4.5, -9), (0, 4, 0)),
    "tree": ("lookat", (0, 2.2, -4), (0, 1, 1)),
}
FIELDS = ["stime", "width", "height", "iter_count", "bounce_count", "ray_count", "light_count", "range", "max_cost_default",
          "debug_nx", "debug_ny", "debug_nz", "debug_scale", "debug_x", "debug_y", "debug_z", "show_objects"]


def frame_to_dict(f):
    d = {k: getattr(f, k) for k in FIELDS}
    d["basis"] = np.array([list(f.eye), list(f.front), list(f.right), list(f.top)], np.float32)
    d["scene_var"] = np.array(list(f.scene_var), np.float32)
    return d


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for scene, (kind, eye, tgt) in CAMS.items():
        basis = (po.camera_lookat if kind == "lookat" else po.camera_direction)(eye, tgt, FOVY, np.float32(W) / np.float32(H))
        for stime in (0.0, 1.25):
            f = po.default_frame(scene, W, H, basis=basis, stime=stime)
            img, st, tot = po.render(scene, f, stats=True)
            np.savez_compressed(os.path.join(out_dir, "%s_t%03d.npz" % (scene, int(stime * 100))), rgba=img, stats=st.astype(np.uint16),
                                totals=tot, eye=np.array(eye, np.float64), target=np.array(tgt, np.float64), target_is_direction=(kind == "dir"),
                                **frame_to_dict(f))
            print(scene, stime, tot)


if __name__ == "__main__":
    main()
