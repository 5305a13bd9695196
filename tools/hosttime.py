import sys, time, os
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py"]
import torch, bench, sdf_playground_amd as sp
r = sp.SDFRenderer(0); r.initShader("labyrinth"); r.setLimits(iter_count=256)
W, H = 256, 144
img = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
def step(s):
    cam, stime = bench.make_camera(s % 16, W, H)
    r.setParameters(stime); r.setCamera(cam); r.render(None, W, H, out=img)
for s in range(20): step(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(500): step(s)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per step: %.1f us; with drain %.1f us" % ((t1 - t0) / 500 * 1e6, (t2 - t0) / 500 * 1e6))
t0 = time.perf_counter()
for s in range(500): bench.make_camera(s % 16, W, H)
print("make_camera %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6))
cam, st = bench.make_camera(3, W, H)
t0 = time.perf_counter()
for s in range(500): r.setCamera(cam)
print("setCamera %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6))
t0 = time.perf_counter()
for s in range(500): r.render(None, W, H, out=img)
print("render enqueue %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6))
torch.cuda.synchronize()
