"""Experiment: is q' = fma(fma(-c, a*rc, a), rc, a*rc), rc = RN(1/c), the correctly rounded a/c
for ARBITRARY divisors c (not only the scene constants)?  Sweeps random and adversarial
divisors, each against all 2^32 numerators in range, on the GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdf_playground_amd as sp
r = sp.SDFRenderer(0)
rng = np.random.default_rng(1)
cs = []
cs += list(np.exp(rng.uniform(np.log(1e-20), np.log(2.0), 150)).astype(np.float32))      # ground-plane denominators
cs += list(rng.uniform(1e-6, 1.0, 100).astype(np.float32))
# adversarial significands: all ones, one below/above powers of two, alternating bits
for e in (-60, -20, -3, -1, 0):
    for m in (0x7FFFFF, 0x7FFFFE, 0x000001, 0x000000, 0x555555, 0x2AAAAA, 0x400000, 0x3FFFFF, 0x400001):
        cs.append(np.uint32(((127 + e) << 23) | m).view(np.float32))
bad = []
for i, c in enumerate(cs):
    n = r.selftestMath(3, float(c))
    if n:
        bad.append((float(c), hex(int(np.float32(c).view(np.uint32))), n))
print("divisors tested", len(cs), "with mismatches", len(bad))
for b in bad[:40]:
    print(b)
print("sqrt selftest", r.selftestMath(0))
