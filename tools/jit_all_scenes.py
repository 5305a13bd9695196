"""One-off check: every built-in scene, cut out of its header and compiled at run time (hiprtc),
renders the same bits as its ahead-of-time build (pixel schedule, 160x100, default camera)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sdf_playground_amd as sp
from jit_util import CSRC

structs = {}
for fn in ("sdfr_scenes.h", "sdfr_scenes2.h", "sdfr_scenes3.h", "sdfr_scenes4.h"):
    text = open(os.path.join(CSRC, fn)).read()
    for m in re.finditer(r"^struct (Scene\w+)\n\{\n.*?^\};\n", text, re.S | re.M):
        name = re.search(r'name\(\) \{ return "(\w+)"', m.group(0)).group(1)
        structs[name] = re.sub(r"\b%s\b" % m.group(1), "Scene", m.group(0))
r = sp.SDFRenderer(0)
cam = sp.Camera(); cam.SetAspect(1.6)
bad = 0
for scene in sp.scene_names():
    r.initShader(scene); r.setParameters(0.7)
    a, sa = r.render(cam, 160, 100, pixel_stats=True)
    try:
        r.initShaderSource(scene + "_rt", structs[scene])
    except sp.SdfrError as e:
        print(scene, "DOES NOT COMPILE:", str(e)[:300]); bad += 1; continue
    r.setParameters(0.7)
    b, sb = r.render(cam, 160, 100, pixel_stats=True)
    ok = np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(sa, sb)
    print(scene, "ok" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
print("failures:", bad)
