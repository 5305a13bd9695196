"""Developer timing of the post-processing kernels (bloom + tone map) at 4K, per kernel (named GPU timings of the
handle): GB/s against the algorithmic bytes (8+8 per pixel for the horizontal pass, 8+8+4 for vertical + tone map),
for frames with little bloom (labyrinth, fast_sphere), with much (light_shadows) and for a dense random image
(every block runs its taps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import sdf_playground_amd as sp
from quickbench import camera_for

W, H = 3840, 2160
r = sp.SDFRenderer(0)
r.setStream(torch.cuda.current_stream().cuda_stream)
hdr = sp.HDR(r)
hdr.init(W, H)


def measure(label):
    for _ in range(3):
        hdr.process()
    t = {"Bloom 1": [], "Bloom 2 + HDR": []}
    for _ in range(10):
        hdr.process()
        got = r.getTimings()
        for k in t:
            t[k].append(got[k])
    b1, b2 = float(np.median(t["Bloom 1"])), float(np.median(t["Bloom 2 + HDR"]))
    lit = float((hdr._bloom != 0).any(dim=2).float().mean())
    print("%-28s bloom_h %6.1f us (%4.2f TB/s, %4.1f %% of 8 TB/s)   bloom_v+tone %6.1f us (%4.2f TB/s, %4.1f %%)   total %6.1f us; %.1f %% of bloom1 texels non-zero" % (
        label, b1 * 1e3, 16.0 * W * H / b1 / 1e9, 16.0 * W * H / b1 / 1e9 / 8 * 100, b2 * 1e3, 20.0 * W * H / b2 / 1e9, 20.0 * W * H / b2 / 1e9 / 8 * 100,
        (b1 + b2) * 1e3, 100 * lit), flush=True)


for scene in ("labyrinth", "fast_sphere", "cube_sea", "light_shadows"):
    r.initShader(scene)
    r.setLimits(iter_count=256)
    r.render(camera_for(scene, 3, W, H), W, H, out=hdr.getRenderTarget(), fmt=sp.RGBA16F)
    measure("frame of " + scene)
g = torch.Generator(device="cuda").manual_seed(1)
hdr.getRenderTarget().copy_((torch.rand((H, W, 4), generator=g, device="cuda") ** 3 * 6).to(torch.float16))
measure("dense random image")
hdr.getRenderTarget().zero_()
measure("black image")
r.close()
