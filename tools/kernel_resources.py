"""Developer tool + the CPU tier's resource guard (tests/test_kernel_resources_cpu.py): what the compiler made of the pixel kernels
of the seven BASELINE configuration scenes, with the options the build ships them with (buildlib.FLAGS + SCENE_FLAGS).

    python tools/kernel_resources.py            prints the table
    python tools/kernel_resources.py --write    and records it as tests/golden/kernel_resources.json (the guard's baseline)

Per scene: vector registers against the scene's budget (512 / waves_per_simd, in steps of 8), scratch bytes per lane, and for the
march loops (tools/isa_loops.py: march_loops) the scratch instructions (must be none) and the three-source fma / fmac whose sources
all lie in one VGPR bank -- half rate on gfx950 (profiles/r03_bank_ubench.txt), invisible to the register allocator, and re-rolled
by unrelated edits: the guard exists so that such a re-roll fails a test instead of costing 3 % unnoticed."""
import json
import os
import re
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import isa_loops  # noqa: E402

CONFIG_SCENES = ["SceneFastSphere", "SceneCubeSea", "SceneLabyrinth", "SceneFractal", "SceneLense", "SceneGems", "SceneLightShadows"]
BASELINE = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")


def waves_per_simd(scene):
    """the scene's `waves_per_simd` declaration (default 7: sdfr_pixel_kernel.h PixelWavesPerSimd)"""
    for f in os.listdir(isa_loops.CSRC):
        if f.startswith("sdfr_scene"):
            text = open(os.path.join(isa_loops.CSRC, f)).read()
            m = re.search(r"struct %s\b(.*?)\n};" % scene, text, re.S)
            if m:
                w = re.search(r"static constexpr int waves_per_simd = (\d+);", m.group(1))
                return int(w.group(1)) if w else 7
    raise KeyError(scene)


def vgpr_budget(waves):
    return (512 // waves) // 8 * 8


def analyse(scene, out_dir="/tmp"):
    asm, remarks = isa_loops.compile_scene(scene, out_dir=out_dir)
    res = isa_loops.kernel_resources(remarks, scene)
    loops, whole, _n = isa_loops.kernel_loops(asm, scene)
    ml = isa_loops.march_loops(loops)
    w = waves_per_simd(scene)
    return {
        "waves_per_simd": w, "vgpr_budget": vgpr_budget(w), "vgprs": res["vgprs"], "sgprs": res["sgprs"], "scratch_bytes": res["scratch_bytes"],
        "occupancy": res["occupancy"], "lds_bytes": res["lds_bytes"],
        "march_loops": [{"valu": s["valu"], "fma": s["fma"], "one_bank_fma": s["one_bank"], "scratch": s["scratch"]} for _a, _b, s in ml],
        "march_loop_scratch": sum(s["scratch"] for _a, _b, s in ml),
        "march_loop_one_bank_fma": sum(s["one_bank"] for _a, _b, s in ml),
        "march_loop_fma": sum(s["fma"] for _a, _b, s in ml),
        "kernel_one_bank_fma": whole["one_bank"], "kernel_valu": whole["valu"],
    }


def table(out_dir="/tmp", jobs=None):
    with ThreadPoolExecutor(max_workers=jobs or min(7, os.cpu_count() or 1)) as ex:
        return dict(zip(CONFIG_SCENES, ex.map(lambda s: analyse(s, out_dir), CONFIG_SCENES)))


def main():
    t = table()
    print("%-18s %5s %6s %7s %7s %5s   march loops: valu / fma / one-bank fma / scratch" % ("scene", "waves", "VGPRs", "budget", "scratch", "occ"))
    for name, r in t.items():
        print("%-18s %5d %6d %7d %7d %5d   %s" % (name, r["waves_per_simd"], r["vgprs"], r["vgpr_budget"], r["scratch_bytes"], r["occupancy"],
                                                  "  ".join("%d/%d/%d/%d" % (m["valu"], m["fma"], m["one_bank_fma"], m["scratch"]) for m in r["march_loops"])))
    if "--write" in sys.argv:
        with open(BASELINE, "w") as f:
            json.dump(t, f, indent=1, sort_keys=True)
            f.write("\n")
        print("wrote", BASELINE)


if __name__ == "__main__":
    main()
