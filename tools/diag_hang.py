import sys, time, faulthandler
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, '.')
import numpy as np
import sdf_playground_amd as sp
r = sp.SDFRenderer(0)
for scene in ("labyrinth", "cube_sea", "lense", "terrain", "fast_sphere"):
    r.initShader(scene)
    for size in (64, 96, 200):
        t = time.time()
        img = r.render(sp.Camera(), size, size)
        print(scene, size, "ok %.3f s" % (time.time() - t), float(np.nansum(img)), flush=True)
print("done", flush=True)
