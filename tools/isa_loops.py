"""Developer tool: instruction mix of the loops of a pixel kernel (compiler's ISA, gfx950).

    python tools/isa_loops.py <scene struct name, e.g. SceneCubeSea> [extra hipcc flags]

Compiles the scene's group of csrc/sdfr_kernels_group.hip to assembly and lists every loop of
k_pixel<Scene, false> with its instruction counts by issue class (tools/ubench: full rate /
half rate / transcendental), spills and memory operations."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdf_playground_amd", "csrc")
HALF = ("v_min", "v_max", "v_med3", "v_floor", "v_trunc", "v_rndne", "v_fract", "v_cvt", "v_lshl", "v_lshr", "v_ashr", "v_cmp", "v_cndmask", "v_and", "v_or", "v_xor",
        "v_bfe", "v_bfi", "v_cmpx", "v_div_", "v_ldexp", "v_frexp", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i", "v_add_co", "v_addc", "v_sub_co", "v_subb", "v_readlane", "v_writelane", "v_readfirstlane")
TRANS = ("v_rsq", "v_rcp", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def compile_scene(scene, extra_flags=None, out_dir="/tmp"):
    """hipcc -S of the scene's compile unit with the shipped options (buildlib.FLAGS + the scene's own, buildlib.SCENE_FLAGS);
    returns (assembly text, the compiler's kernel-resource-usage remarks)."""
    text = open(os.path.join(CSRC, "sdfr_perpixel.h")).read()
    idx = int(re.search(r"X\((\d+), %s\)" % scene, text).group(1))
    groups = int(re.search(r"#define SDFR_GROUPS (\d+)", text).group(1))
    out = os.path.join(out_dir, "isa_%s.s" % scene)
    sys.path.insert(0, ROOT)
    from sdf_playground_amd.buildlib import group_flags  # the scene's own code-generation options, unless the caller passes some
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-x", "hip", "-Wno-unused-result",
           "-Wno-unknown-pragmas", "-I" + CSRC, "-DSDFR_GROUP=%d" % (idx % groups), "--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage",
           os.path.join(CSRC, "sdfr_kernels_group.hip"), "-o", out] + (list(extra_flags) if extra_flags else group_flags(idx % groups))
    r = subprocess.run(cmd, check=True, stderr=subprocess.PIPE, text=True)
    return open(out).read(), r.stderr


def kernel_resources(remarks, scene, dbg=False):
    """{VGPRs, SGPRs, ScratchSize, Occupancy, LDS} of k_pixel<Scene, dbg> from -Rpass-analysis=kernel-resource-usage"""
    want = re.compile(r"Function Name: _ZN4sdfr7k_pixelINS_\d+%sELb%d" % (scene, 1 if dbg else 0))
    lines = remarks.split("\n")
    for i, l in enumerate(lines):
        if want.search(l):
            out = {}
            for m in lines[i + 1:i + 14]:
                for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("sgprs", r"SGPRs: (\d+)"), ("scratch_bytes", r"ScratchSize \[bytes/lane\]: (\d+)"),
                                 ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"),
                                 ("lds_bytes", r"LDS Size \[bytes/block\]: (\d+)")):
                    mm = re.search(pat, m)
                    if mm and key not in out:
                        out[key] = int(mm.group(1))
            return out
    raise KeyError(scene)


def one_bank(l):
    """a three-source instruction (v_fma_f32; v_fmac_f32, whose destination is the addend) with all three vector sources in one
    register bank (number mod 4) issues at half rate: tools/ubench/bank_ubench.hip, profiles/r03_bank_ubench.txt"""
    m = re.match(r"(v_\w+)\s+(.*)", l.split(";")[0])
    if not m:
        return False
    ops = [o.strip() for o in m.group(2).split(",")]
    srcs = ops if m.group(1).startswith(("v_fmac", "v_mac")) else ops[1:]
    regs = [int(x.group(1)) for x in (re.match(r"^[-|]*v(\d+)\|?$", o) for o in srcs) if x]
    return len(regs) >= 3 and len(set(r % 4 for r in regs)) == 1


def kernel_loops(asm, scene):
    """[(first line, last line, stats)] of every loop of k_pixel<Scene, false>, and the whole kernel's stats"""
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4sdfr7k_pixelINS_\d+%sELb0.*:" % scene, l))
    end = start
    while not lines[end].startswith(".Lfunc_end"):
        end += 1
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = set()
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.add((labels[m.group(1)], i))

    def stats(a, b):
        ins = [l.strip() for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
        valu = [l for l in ins if l.startswith("v_")]
        sgpr_operand = sum(1 for l in valu if not l.startswith(HALF + TRANS) and re.search(r"[, ]s\d+|[, ]s\[", l.split(";")[0]))
        return dict(n=len(ins), valu=len(valu), half=sum(1 for l in valu if l.startswith(HALF)), trans=sum(1 for l in valu if l.startswith(TRANS)),
                    sgpr_op=sgpr_operand, one_bank=sum(1 for l in valu if one_bank(l)), fma=sum(1 for l in valu if l.startswith(("v_fma_f32", "v_fmac_f32"))), salu=sum(1 for l in ins if l.startswith("s_")), scratch=sum(1 for l in ins if l.startswith("scratch_")),
                    lds=sum(1 for l in ins if l.startswith("ds_")), vmem=sum(1 for l in ins if l.startswith(("global_", "buffer_", "flat_"))))

    return [(a, b, stats(a, b)) for a, b in sorted(loops, key=lambda x: x[1] - x[0])], stats(0, len(body) - 1), len(body)


def march_loops(loops):
    """The loops a ray spends its life in: loops that evaluate the scene (at least 20 fma / fmac: a scene evaluation is made of
    them, the tile and queue loops around have none) and contain no other such loop -- the march loop of each code path."""
    heavy = [(a, b, s) for a, b, s in loops if s["fma"] >= 20]
    return [(a, b, s) for a, b, s in heavy if not any(a <= c and d <= b and (c, d) != (a, b) for c, d, _t in heavy)]


def main():
    scene = sys.argv[1]
    asm, remarks = compile_scene(scene, sys.argv[2:] or None)
    loops, whole, n = kernel_loops(asm, scene)
    print("%s: kernel %d lines; %s" % (scene, n, kernel_resources(remarks, scene)))
    print("  lines          insts  valu  (half-rate  trans  full+sgpr-operand)  salu scratch lds vmem   fma/fmac: three sources in one bank")
    inner = {(a, b) for a, b, _s in march_loops(loops)}
    for a, b, s in loops:
        if s["valu"] < 40:
            continue
        print("  %5d-%5d  %6d %5d  (%5d %6d %6d)  %13d %5d %4d %4d   %4d: %d%s" % (a, b, s["n"], s["valu"], s["half"], s["trans"], s["sgpr_op"], s["salu"], s["scratch"], s["lds"], s["vmem"], s["fma"], s["one_bank"], "   <- march loop" if (a, b) in inner else ""))
    s = whole
    print("  whole kernel   %6d %5d  (%5d %6d %6d)  %13d %5d %4d %4d   %4d: %d" % (s["n"], s["valu"], s["half"], s["trans"], s["sgpr_op"], s["salu"], s["scratch"], s["lds"], s["vmem"], s["fma"], s["one_bank"]))


if __name__ == "__main__":
    main()
