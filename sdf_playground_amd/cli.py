"""Headless front end of the renderer: what the reference's scene picker
(SceneManager), slider panel (VariableManager) and window do interactively, as a command.

    python -m sdf_playground_amd.cli --list-scenes
    python -m sdf_playground_amd.cli --scene lense --list-vars
    python -m sdf_playground_amd.cli --scene lense --set mixing=0.8 --set zpos=9 --time 1.5 \\
            --size 1200x800 --eye 0,0.5,7 --lookat 0,0,0 --out lense.png
    python -m sdf_playground_amd.cli --parse-hlsl path/to/sdf_scene_x.hlsl      # VAR_ tags of a scene file
    python -m sdf_playground_amd.cli --scene-source sdf_playground_amd/scenes/pendulum.scene.h --out p.png
    python -m sdf_playground_amd.cli --scene-source my.scene.h --check              # compile only, no GPU
    python -m sdf_playground_amd.cli --scene-hlsl sdf_playground_amd/scenes/pendulum.hlsl --out p.png   # a scene in the reference's dialect
    python -m sdf_playground_amd.cli --scene-hlsl Engine/shader/scenes/sdf_scene_tree.hlsl --translate   # the generated C++

--out writes the tone-mapped + bloomed LDR image (HDR::process, like the reference's window);
--out-hdr writes the raw float32 RGBA frame as .npy.  Needs a GPU.
"""
import argparse
import struct
import sys
import zlib

import numpy as np


def write_png(path, rgba8):
    h, w, _ = rgba8.shape
    raw = b"".join(b"\x00" + rgba8[y, :, :3].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def parse_var_tags(text):
    """The reference's VAR_ tag rules (ShaderUtil.cpp:122-191) in Python, for scene *files*:
    name -> (min, max, start, step).  The library's own parser is the C++ one."""
    out = {}
    pos = 0
    while True:
        a = text.find("VAR_", pos)
        if a < 0:
            break
        b = text.find(")", a + 4)
        if b < 0:
            break
        tag = text[a:b + 1]
        pos = b + 1
        lb = tag.find("(")
        name = tag[4:lb]
        kv = {}
        for part in tag[lb + 1:-1].split(","):
            sides = part.split("=")
            if len(sides) != 2:
                break
            kv[sides[0].split()[0] if sides[0].split() else ""] = float(np.float32(float(sides[1].split()[0])))
        mn = kv.get("min", 0.0)
        mx = kv.get("max", 2.0)
        start = kv.get("start", float(np.float32(np.float32(mx + mn) * np.float32(0.5))))
        step = kv.get("step", float(np.float32(np.float32(mx - mn) * np.float32(0.05))))
        out[name] = (mn, mx, start, step)
    return dict(sorted(out.items()))


def _vec(s):
    v = tuple(float(x) for x in s.split(","))
    if len(v) != 3:
        raise argparse.ArgumentTypeError("expected x,y,z")
    return v


def main(argv=None):
    ap = argparse.ArgumentParser(prog="sdf_playground_amd.cli", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--list-scenes", action="store_true")
    ap.add_argument("--scene")
    ap.add_argument("--scene-source", metavar="FILE", help="scene compiled at run time (scenes/README.md)")
    ap.add_argument("--scene-hlsl", metavar="FILE", help="scene in the reference's own dialect: an .hlsl scene file with map / map_normal / map_light / "
                                                         "map_background (sdfr_load_scene_hlsl)")
    ap.add_argument("--translate", action="store_true", help="with --scene-hlsl: print the C++ generated from the file and stop")
    ap.add_argument("--check", action="store_true", help="with --scene-source / --scene-hlsl: compile only, no GPU needed")
    ap.add_argument("--list-vars", action="store_true")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE")
    ap.add_argument("--time", type=float, default=0.0)
    ap.add_argument("--size", default="1200x800")
    ap.add_argument("--eye", type=_vec, default=(0.0, 2.0, -3.0))
    ap.add_argument("--lookat", type=_vec, default=None)
    ap.add_argument("--direction", type=_vec, default=None)
    ap.add_argument("--fovy", type=float, default=60.0, help="degrees")
    ap.add_argument("--roll", type=float, default=0.0, help="radians")
    ap.add_argument("--iter-count", type=int, default=100)
    ap.add_argument("--max-cost", type=int, default=7)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out")
    ap.add_argument("--out-hdr")
    ap.add_argument("--parse-hlsl", metavar="FILE")
    a = ap.parse_args(argv)

    if a.parse_hlsl:
        for name, (mn, mx, start, step) in parse_var_tags(open(a.parse_hlsl).read()).items():
            print("%-16s min %-8g max %-8g start %-8g step %g" % (name, mn, mx, start, step))
        return 0

    import sdf_playground_amd as sp

    if a.list_scenes:
        print("\n".join(sp.scene_names()))
        return 0
    if a.scene_source and a.check:
        ok, log = sp.check_scene_source(a.scene_source)
        print("ok" if ok else log)
        return 0 if ok else 1
    if a.scene_hlsl and a.translate:
        print(sp.translate_scene_hlsl(a.scene_hlsl))
        return 0
    if a.scene_hlsl and a.check:
        ok, log = sp.check_scene_hlsl(a.scene_hlsl)
        print("ok" if ok else log)
        return 0 if ok else 1
    if not a.scene and not a.scene_source and not a.scene_hlsl:
        ap.error("--scene, --scene-source or --scene-hlsl is required")
    r = sp.SDFRenderer(a.device)
    if a.scene_source or a.scene_hlsl:
        import os

        path = a.scene_source or a.scene_hlsl
        a.scene = os.path.basename(path).split(".")[0]
        try:
            (r.initShaderSource if a.scene_source else r.initShaderHlsl)(a.scene, path)
        except sp.SdfrError as e:
            print(e, file=sys.stderr)
            return 1
    else:
        r.initShader(a.scene)
    for item in a.set:
        name, _, value = item.partition("=")
        if not r.setValue(name, float(value)):
            print("warning: unknown variable %r ignored (ShaderVariableManager::setValue)" % name, file=sys.stderr)
    if a.list_vars:
        for v in r.getVariableMap().values():
            print("%-16s min %-8g max %-8g start %-8g step %-8g value %g" % (v.name, v.minval, v.maxval, v.start, v.step, v.value))
        return 0
    w, h = (int(x) for x in a.size.lower().split("x"))
    cam = sp.Camera()
    cam.SetEye(a.eye)
    if a.direction is not None:
        cam.SetDirection(a.direction)
    else:
        cam.SetLookat(a.lookat if a.lookat is not None else (0.0, 1.0, 0.0))
    cam.SetFOVY(sp.to_radian(a.fovy))
    cam.SetAspect(float(np.float32(w) / np.float32(h)))
    cam.SetRoll(a.roll)
    r.setLimits(iter_count=a.iter_count, max_cost_default=a.max_cost)
    r.setParameters(a.time)
    if a.out_hdr:
        np.save(a.out_hdr, r.render(cam, w, h))
    if a.out:
        hdr = sp.HDR(r)
        hdr.init(w, h)
        r.render(cam, w, h, out=hdr.getRenderTarget(), fmt=sp.RGBA16F)
        write_png(a.out, hdr.process().cpu().numpy())
    s = r.getStats()
    print("%s %dx%d: %.3f ms, %d rays (%.1f Mrays/s)" % (a.scene, w, h, s.ms_gpu, s.rays, s.rays / max(s.ms_gpu, 1e-9) / 1e3))
    print("  ".join("%s %.3f ms" % kv for kv in r.getTimings().items()))
    r.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
