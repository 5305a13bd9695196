"""Developer timing table: every built-in scene at 3840x2160, iter_count 256, pixel schedule
(median of 3 frames after a warm-up), plus the host-destination (PCIe-inclusive) labyrinth rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import sdf_playground_amd as sp
from quickbench import camera_for

W, H = 3840, 2160
ONLY = sys.argv[sys.argv.index("--scenes") + 1:] if "--scenes" in sys.argv else None  # a few scenes only (A/B of one scene's change)
r = sp.SDFRenderer(0)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
print("scene                 ms/frame   Mrays/s  rays/px  evals/ray")
for scene in (ONLY or sp.scene_names()):
    r.initShader(scene)
    r.setLimits(iter_count=256)
    ms, st = [], None
    for k in range(4):
        r.setParameters(k / 60.0)
        r.render(camera_for(scene, k, W, H), W, H, out=out)
        st = r.getStats()
        if k:
            ms.append(st.ms_gpu)
    m = float(np.median(ms))
    print("%-20s %9.3f %9.1f %8.2f %10.1f" % (scene, m, st.rays / m / 1e3, st.rays / st.pixels, st.march_evals / max(1, st.rays)), flush=True)
if ONLY:
    r.close()
    sys.exit(0)
r.initShader("labyrinth")
r.setLimits(iter_count=256)
cam = camera_for("labyrinth", 1, W, H)
r.render(cam, W, H)
t0 = time.perf_counter()
for _ in range(5):
    img = r.render(cam, W, H)  # a fresh numpy array every frame: kernel + 132.7 MB device-to-host copy through pageable memory
dt = (time.perf_counter() - t0) / 5
print("labyrinth to a host buffer (PCIe-inclusive), a new buffer every frame: %.2f ms/frame, %.0f Mrays/s" % (dt * 1e3, r.getStats().rays / dt / 1e6))
host = np.zeros((H, W, 4), np.float32)
r.registerHostTarget(host)  # page-locked once: the host's persistent render target
r.render(cam, W, H, out=host)
t0 = time.perf_counter()
for _ in range(5):
    r.render(cam, W, H, out=host)  # the same buffer every frame, as a host with one render target does
dt = (time.perf_counter() - t0) / 5
assert np.array_equal(host, img)
r.registerHostTarget(None)
print("labyrinth to a registered host buffer (PCIe-inclusive, sdfr_register_host_target): %.2f ms/frame, %.0f Mrays/s" % (dt * 1e3, r.getStats().rays / dt / 1e6))
r.close()
