"""Developer timing of the post-processing kernels (bloom + tone map) at 4K: GB/s against the
algorithmic 36 B/pixel (8+8 for the horizontal pass, 8+8+4 for vertical + tone map)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdf_playground_amd as sp

W, H = 3840, 2160
r = sp.SDFRenderer(0)
r.setStream(torch.cuda.current_stream().cuda_stream)
r.initShader("light_shadows")
hdr = sp.HDR(r)
hdr.init(W, H)
cam = sp.Camera()
cam.SetEye((0, 5, -9)); cam.SetLookat((0, 1, 0)); cam.SetAspect(W / H)
r.render(cam, W, H, out=hdr.getRenderTarget(), fmt=sp.RGBA16F)
for _ in range(3):
    hdr.process()
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    hdr.process()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("postprocess 3840x2160: %.3f ms/frame, %.1f GB/s algorithmic (36 B/px), %.1f %% of 8 TB/s" % (ms, 36.0 * W * H / ms / 1e6, 36.0 * W * H / ms / 1e6 / 80.0))
