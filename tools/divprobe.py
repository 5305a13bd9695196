import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_playground_amd as sp
r = sp.SDFRenderer(0)
for c in [float(x) for x in sys.argv[1:]]:
    print(c, r.selftestMath(1, c), flush=True)
