"""Builds libsdfr.so (C ABI + gfx950 kernels) in-tree with hipcc.

Flags that matter for correctness:
  -ffp-contract=off   only the explicit fma() calls fuse (arithmetic contract)
  -fno-slp-vectorize  no v_pk_*_f32: on gfx950 a packed fp32 op costs more issue time than the two
                      scalar ops it replaces (tools/ubench: pk_add 6.1 vs 2 x 2.5 cycles); measured at
                      4K: labyrinth -4 %, cube_sea -15 %, lense +5 %
  default hipcc fp32 divide/sqrt are correctly rounded and denormals are kept; do not add
  -ffast-math / -fgpu-flush-denormals-to-zero / -fno-hip-fp32-correctly-rounded-divide-sqrt.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsdfr.so")
SOURCES = ["sdfr_api.cpp", "sdfr_comm.cpp", "sdfr_jit.cpp", "sdfr_kernels.hip", "sdfr_post.hip"]
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "sdfr.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=(), out=None):
    """Compile every HIP source for gfx950 into sdf_playground_amd/libsdfr.so (or `out`)."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-std=c++17", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-x", "hip",
           "-Wno-unused-result", "-Wno-unknown-pragmas", "-I" + CSRC] + list(extra)
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out or LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out or LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
