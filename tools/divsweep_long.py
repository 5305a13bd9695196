"""Long sweep behind the per-ray reciprocal divisions (ground plane, cube_sea cell guard): for many
random divisors c (all significand patterns equally likely, exponents 2^-66 .. 2^1), is
div_c(a, c, RN(1/c)) the correctly rounded a / c for EVERY numerator a = 0 or 2^-60 <= |a| <= 2^40?
(sdfr_selftest_math what = 3, exhaustive over numerators on the GPU.)  Prints progress."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdf_playground_amd as sp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
r = sp.SDFRenderer(0)
rng = np.random.default_rng(seed)
bits = ((rng.integers(61, 129, n).astype(np.uint32)) << 23) | rng.integers(0, 1 << 23, n).astype(np.uint32)
cs = bits.view(np.float32)
bad = []
t0 = time.time()
for i, c in enumerate(cs):
    m = r.selftestMath(3, float(c))
    if m:
        bad.append((float(c), hex(int(bits[i])), m))
        print("MISMATCH", bad[-1], flush=True)
    if i % 200 == 199:
        print("%d divisors, %d with mismatches, %.0f s" % (i + 1, len(bad), time.time() - t0), flush=True)
print("divisors tested", n, "with mismatches", len(bad))
