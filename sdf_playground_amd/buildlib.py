"""Builds libsdfr.so (C ABI + gfx950 kernels) in-tree with hipcc.

Every source is compiled to an object of its own, in parallel; the per-scene kernels
(csrc/sdfr_kernels_group.hip) are compiled once per scene (-DSDFR_GROUP=<scene index>, SDFR_GROUPS units), so that
the build takes ~20 s on 8 cores instead of a minute and a scene can have code-generation options of its own
(SCENE_FLAGS below).

Flags that matter for correctness:
  -ffp-contract=off   only the explicit fma() calls fuse (arithmetic contract)
  -fno-slp-vectorize  no v_pk_*_f32: on gfx950 a packed fp32 op costs more issue time than the two
                      scalar ops it replaces (tools/ubench: pk_add 6.1 vs 2 x 2.5 cycles); measured at
                      4K: labyrinth -4 %, cube_sea -15 %, lense +5 %
  default hipcc fp32 divide/sqrt are correctly rounded and denormals are kept; do not add
  -ffast-math / -fgpu-flush-denormals-to-zero / -fno-hip-fp32-correctly-rounded-divide-sqrt.
"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsdfr.so")
OBJDIR = os.path.join(HERE, "build")
SOURCES = ["sdfr_api.cpp", "sdfr_comm.cpp", "sdfr_hlsl.cpp", "sdfr_jit.cpp", "sdfr_kernels.hip", "sdfr_post.hip"]
GROUP_SOURCE = "sdfr_kernels_group.hip"
ARCH = "gfx950"
# Per scene: options that change the register ASSIGNMENT or the instruction ORDER of its kernels, never the arithmetic.  A three-source
# instruction (v_fma_f32, v_fmac_f32) whose sources all lie in one VGPR bank (register number mod 4) issues at half rate on gfx950
# (tools/ubench/bank_ubench.hip, profiles/r03_bank_ubench.txt), the allocator does not know, and how many of a march loop's fma land
# that way is luck: tools/isa_loops.py counts them, and a scene whose default draw is a bad one gets another (each scene is a compile
# unit of its own).  Chosen by that count, kept where the GPU confirmed it (profiles/r03_launch_experiments.txt):
#   fractal 15 -> 4 of its march loop's 104 fma, coordinate_material, terrain (-2.0 / -1.1 / -1.9 % at 4K), tiling (-4 %): local live ranges
#   assigned shortest first; cube_sea 4 -> 2 of 33 (BASELINE configuration 2 -2.1 %), shell (-2 %): scheduler strategy max-memory-clause;
#   lense 12 -> 0 of 79 (configuration 5 -0.9 %): top-down pre-RA scheduling (which costs distortion 12 %, hence per scene).
_REV = ["-mllvm", "-greedy-reverse-local-assignment"]
_MAXMEM = ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]
_TOPDOWN = ["-mllvm", "-misched-prera-direction=topdown"]
_MAXILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
# round 4 (tools/flag_search.py, gpurun_out/r04/s14): the fractal's 104-fma march loop draws 1 one-bank fma under max-ilp (4 under the reversed
# local assignment it had, 15 by default): BASELINE configuration 4 1.044 / 1.048 -> 1.031 / 1.025 ms (-1.7 %), 20 bytes more scratch
# outside the loop.  The same search left the labyrinth (default draw: rev +0.8 %, top-down +3 %), gems (+-0) and cube_sea where they were.
SCENE_FLAGS = {"SceneFractal": _MAXILP, "SceneCoordinateMaterial": _REV, "SceneTerrain": _REV, "SceneTiling": _REV,
               "SceneCubeSea": _MAXMEM, "SceneShell": _MAXMEM, "SceneLense": _TOPDOWN}


def scene_registry():
    """[(index, struct name)] of the scenes compiled ahead of time (csrc/sdfr_perpixel.h)."""
    text = open(os.path.join(CSRC, "sdfr_perpixel.h")).read()
    return [(int(i), n) for i, n in re.findall(r"X\((\d+), (Scene\w+)\)", text)]


def group_flags(g, override=None):
    """Options of compile unit g: those of the scenes in it (one scene per unit: SDFR_GROUPS >= scene count).
    override: {scene struct name: [flags]} replaces SCENE_FLAGS for those scenes (developer A/B of another register draw)."""
    groups = scene_groups()
    flags = []
    table = dict(SCENE_FLAGS)
    table.update(override or {})
    for i, name in scene_registry():
        if i % groups == g:
            flags += [f for f in table.get(name, []) if f not in flags or f == "-mllvm"]
    return flags


FLAGS = ["--offload-arch=" + ARCH, "-std=c++17", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-x", "hip", "-Wno-unused-result",
         "-Wno-unknown-pragmas", "-I" + CSRC]


def scene_groups():
    text = open(os.path.join(CSRC, "sdfr_perpixel.h")).read()
    return int(re.search(r"#define SDFR_GROUPS (\d+)", text).group(1))


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "sdfr.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False, extra=(), out=None, jobs=None, scene_flags=None):
    """Compile every HIP source for gfx950 into sdf_playground_amd/libsdfr.so (or `out`).
    scene_flags: {scene struct name: [flags]} instead of SCENE_FLAGS for those scenes (only their units differ from the default build)."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = list(extra)
    objdir = os.path.join(OBJDIR, hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10] if extra else "default")
    os.makedirs(objdir, exist_ok=True)
    units = [(s, [], os.path.join(objdir, s + ".o")) for s in SOURCES]
    for g in range(scene_groups()):
        flags = group_flags(g, scene_flags)
        tag = "" if flags == group_flags(g) else "." + hashlib.sha1(" ".join(flags).encode()).hexdigest()[:8]
        units.append((GROUP_SOURCE, ["-DSDFR_GROUP=%d" % g] + flags, os.path.join(objdir, "%s.%d%s.o" % (GROUP_SOURCE, g, tag))))
    newest_dep = max(os.path.getmtime(d) for d in _deps())

    def compile_unit(u):
        src, defs, obj = u
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest_dep:
            return
        cmd = [hipcc] + FLAGS + extra + defs + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    # the group units first: they are the long ones
    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(compile_unit, sorted(units, key=lambda u: u[0] != GROUP_SOURCE)))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-fPIC", "-shared"] + [u[2] for u in units] + ["-o", out or LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return out or LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
