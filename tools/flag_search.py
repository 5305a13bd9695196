"""Developer tool: other register draws for one scene's kernels.  Builds libsdfr with candidate code-generation options for ONE scene
(options that change register assignment or instruction order, never the arithmetic: buildlib.SCENE_FLAGS) into tools/libsdfr_v<k>.so,
prints what tools/isa_loops.py sees in each (march-loop instructions, one-bank fma, scratch), and writes the A/B command for the GPU box.

    python tools/flag_search.py SceneLabyrinth 3        (scene struct, bench configuration)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_loops
from sdf_playground_amd import buildlib

CANDIDATES = [
    [],
    ["-mllvm", "-greedy-reverse-local-assignment"],
    ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    ["-mllvm", "-misched-prera-direction=topdown"],
    ["-mllvm", "-misched-prera-direction=bottomup"],
    ["-mllvm", "-greedy-reverse-local-assignment", "-mllvm", "-misched-prera-direction=topdown"],
    ["-mllvm", "-greedy-reverse-local-assignment", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    ["-mllvm", "-regalloc-enable-priority-advisor=default", "-mllvm", "-enable-local-reassign"],
    ["-mllvm", "-amdgpu-schedule-relaxed-occupancy"],
    ["-mllvm", "-misched-postra-direction=bottomup"],
    ["-mllvm", "-enable-post-misched=false"],
]


def main():
    scene, config = sys.argv[1], sys.argv[2]
    specs = []
    for k, flags in enumerate(CANDIDATES):
        try:
            asm, rem = isa_loops.compile_scene(scene, flags or ["-Wno-unused-value"])
        except Exception as e:
            print("v%d %s: does not compile (%s)" % (k, " ".join(flags), type(e).__name__))
            continue
        res = isa_loops.kernel_resources(rem, scene)
        loops, whole, _n = isa_loops.kernel_loops(asm, scene)
        ml = isa_loops.march_loops(loops)
        print("v%d %-70s VGPRs %d scratch %d  kernel one-bank %d / valu %d  march loops %s" % (
            k, " ".join(flags) or "(default draw)", res["vgprs"], res["scratch_bytes"], whole["one_bank"], whole["valu"],
            " ".join("%d/%d/%d/%d" % (s["valu"], s["fma"], s["one_bank"], s["scratch"]) for _a, _b, s in ml)), flush=True)
        out = os.path.join(ROOT, "tools", "libsdfr_v%d.so" % k)
        buildlib.build(out=out, scene_flags={scene: flags})
        specs.append('"v%d=SDFR_LIBRARY=$GRAFT_REPO_ROOT/tools/libsdfr_v%d.so"' % (k, k))
    print("\non the GPU box:\n  tools/ab_env.sh %s %s" % (config, " ".join(specs)))


if __name__ == "__main__":
    main()
