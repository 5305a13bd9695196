"""GPU tier: at the cameras and times fitted to the reference's own screenshots (tests/golden/reference_images.json; the fit and the
comparison with the screenshots are CPU-tier work, tests/test_reference_images_cpu.py -- the screenshots do not travel) the HIP path
renders the oracle's bits at the screenshots' size, 1200 x 800: pixels and per-pixel ray / step / hit counters.  Together: screenshot
~ oracle (within the statistics recorded there) and oracle = HIP (exactly)."""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 1200, 800

with open(os.path.join(HERE, "golden", "reference_images.json")) as _f:
    FITS = json.load(_f)["images"]


@pytest.mark.parametrize("name", sorted(FITS))
def test_hip_path_equals_the_oracle_at_the_screenshot_s_camera(oracle, name):
    import sdf_playground_amd as sp

    e = FITS[name]
    scene = "light_shadows" if e["scene"] == "light_shadows_backwards" else e["scene"]  # (the backwards variant is test infrastructure of the oracle)
    direction = (math.cos(e["pitch"]) * math.sin(e["yaw"]), math.sin(e["pitch"]), math.cos(e["pitch"]) * math.cos(e["yaw"]))
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    aspect = np.float32(W) / np.float32(H)
    f = oracle.default_frame(scene, W, H, basis=oracle.camera_direction(e["eye"], direction, fovy, aspect), stime=e["stime"])
    slots = {row[0]: row[6] for row in oracle.var_table(scene)}
    for k, v in e.get("variables", {}).items():  # sliders the screenshot was taken with (neon)
        f.scene_var[slots[k]] = v
    ref, rst, tot = oracle.render(scene, f, stats=True)
    if e["scene"] == scene:
        assert int(tot[1]) == e["stats"]["rays"]  # the frame the statistics were taken on
    r = sp.SDFRenderer(0)
    try:
        r.initShader(scene)
        r.setParameters(e["stime"])
        for k, v in e.get("variables", {}).items():
            assert r.setValue(k, v)
        cam = sp.Camera()
        cam.SetEye(e["eye"])
        cam.SetDirection(direction)
        cam.SetFOVY(float(fovy))
        cam.SetAspect(float(aspect))
        for shortcuts in (False, True):
            r.setStepShortcuts(shortcuts)
            img, st = r.render(cam, W, H, pixel_stats=True)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (name, shortcuts, int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()))
            if shortcuts:
                assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2])
            else:
                assert np.array_equal(st, rst), name
    finally:
        r.close()
