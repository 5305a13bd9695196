"""Generates tests/golden/frames256.json: SHA-256 digests of 256x256 oracle frames (and of their
per-pixel counters) of the seven configuration scenes at the reference's start-up camera
(Application.cpp:214-224 with aspect 1), stime 0 and 1.25, reference limits -- plus BASELINE
config 1 as worded (fast_sphere, 64 steps, max_cost 2: no secondary rays).  Frames of this size
are too large to commit as data for every scene (SURVEY.md 8c asks for them); a digest pins
every bit just as well."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import pyoracle as po

W = H = 256
CASES = [(s, t, {}) for s in ("fast_sphere", "cube_sea", "labyrinth", "fractal", "lense", "gems", "light_shadows") for t in (0.0, 1.25)]
CASES.append(("fast_sphere", 0.0, dict(iter_count=64, max_cost_default=2)))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = []
    for scene, stime, limits in CASES:
        f = po.default_frame(scene, W, H, stime=stime)
        for k, v in limits.items():
            setattr(f, k, v)
        img, st, tot = po.render(scene, f, stats=True)
        out.append({"scene": scene, "stime": stime, "limits": limits, "rgba_sha256": digest(img), "stats_sha256": digest(st),
                    "totals": [int(x) for x in tot]})
        print(out[-1])
    with open(os.path.join(ROOT, "tests", "golden", "frames256.json"), "w") as fh:
        json.dump({"width": W, "height": H, "camera": "start-up: eye (0,2,-3) -> lookat (0,1,0), fovy 60 deg, aspect 1", "frames": out}, fh, indent=1)


if __name__ == "__main__":
    main()
