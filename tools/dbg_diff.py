import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from oracle import pyoracle as po
import sdf_playground_amd as sp
import test_gpu_parity as T
scene = sys.argv[1]; sched = int(sys.argv[2]); stime = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
r = sp.SDFRenderer(0)
f = T._setup(r, po, scene, stime)
r.setSchedule(sched)
img, st = r.render(None, T.W, T.H, pixel_stats=True)
ref, rst, tot = po.render(scene, f, stats=True)
bad = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
print(scene, 'sched', sched, 'bad pixels', bad.sum(), 'of', bad.size, 'stats differ at', (st != rst).any(axis=2).sum())
ys, xs = np.nonzero(bad)
for y, x in list(zip(ys, xs))[:12]:
    print((x, y), 'gpu', img[y, x], 'ref', ref[y, x], 'gpu stats', st[y, x], 'ref stats', rst[y, x])
sb = (st != rst).any(axis=2)
ys, xs = np.nonzero(sb & ~bad)
for y, x in list(zip(ys, xs))[:6]:
    print('stats only', (x, y), st[y, x], rst[y, x])
