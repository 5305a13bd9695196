"""tests/hostsim -- TEST-ONLY CPU build of the product's per-pixel pipeline stages
(sdf_playground_amd/csrc/*.h).  It lets the CPU test tier bit-compare the stage arithmetic
of the HIP kernels with the oracle without a GPU.  The product never loads this library."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "sdf_playground_amd", "csrc")


class FrameU(ctypes.Structure):
    """Mirror of sdfr::FrameU (sdf_playground_amd/csrc/sdfr_frame.h)."""

    _fields_ = [
        ("eye", ctypes.c_float * 3), ("front", ctypes.c_float * 3), ("right", ctypes.c_float * 3), ("top", ctypes.c_float * 3),
        ("stime", ctypes.c_float),
        ("width", ctypes.c_int), ("height", ctypes.c_int),
        ("iter_count", ctypes.c_int), ("bounce_count", ctypes.c_int), ("ray_count", ctypes.c_int), ("light_count", ctypes.c_int),
        ("range", ctypes.c_float), ("max_cost_default", ctypes.c_uint),
        ("debug_nx", ctypes.c_float), ("debug_ny", ctypes.c_float), ("debug_nz", ctypes.c_float), ("debug_scale", ctypes.c_float),
        ("debug_x", ctypes.c_float), ("debug_y", ctypes.c_float), ("debug_z", ctypes.c_float), ("show_objects", ctypes.c_float),
        ("scene_var", ctypes.c_float * 8),
        ("debug_normal", ctypes.c_float * 3), ("debug_plane_on", ctypes.c_int), ("show_on", ctypes.c_int),
        ("ddx", ctypes.c_float), ("ddy", ctypes.c_float), ("sky_s", ctypes.c_float), ("sky_c", ctypes.c_float),
        ("su", ctypes.c_float * 48),
        ("extension_lights", ctypes.c_int), ("ext_light", (ctypes.c_float * 6) * 7), ("extension_marble_reflection", ctypes.c_float),
        ("step_shortcuts", ctypes.c_int),
        ("dist_eps", ctypes.c_float), ("grad_eps", ctypes.c_float), ("reflect_eps", ctypes.c_float), ("refract_eps", ctypes.c_float),
        ("shadow_eps", ctypes.c_float),
        ("widthf", ctypes.c_float), ("heightf", ctypes.c_float),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libhostsim.so")
        srcs = [os.path.join(_HERE, "hostsim.cpp")] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.run(
                ["g++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-mfma", "-mavx2", "-fno-math-errno", "-Wno-unknown-pragmas",
                 "-pthread", "-I" + _CSRC, "-shared", "-o", so, os.path.join(_HERE, "hostsim.cpp")], check=True)
        _lib = ctypes.CDLL(so)
        _lib.hostsim_math.restype = ctypes.c_float
        _lib.hostsim_math.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_float]
        assert _lib.hostsim_frame_size() == ctypes.sizeof(FrameU)
    return _lib


def frame_from_oracle(of):
    """Copy the inputs of an oracle OrcFrame into a product FrameU."""
    f = FrameU()
    lib().hostsim_frame_defaults(ctypes.byref(f))
    for name in ("stime", "width", "height", "iter_count", "bounce_count", "ray_count", "light_count", "range", "max_cost_default",
                 "debug_nx", "debug_ny", "debug_nz", "debug_scale", "debug_x", "debug_y", "debug_z", "show_objects", "extension_lights", "extension_marble_reflection",
                 "dist_eps", "grad_eps", "reflect_eps", "refract_eps", "shadow_eps"):
        setattr(f, name, getattr(of, name))
    for i in range(3):
        f.eye[i], f.front[i], f.right[i], f.top[i] = of.eye[i], of.front[i], of.right[i], of.top[i]
    for i in range(8):
        f.scene_var[i] = of.scene_var[i]
    return f


def render(scene, frame, stats=True, nthreads=None):
    W, H = frame.width, frame.height
    out = np.zeros((H, W, 4), np.float32)
    st = np.zeros((H, W, 3), np.uint32) if stats else None
    rc = lib().hostsim_render(scene.encode(), ctypes.byref(frame), out.ctypes.data_as(ctypes.c_void_p),
                              st.ctypes.data_as(ctypes.c_void_p) if stats else None, nthreads or os.cpu_count() or 1)
    if rc != 0:
        raise ValueError("hostsim_render failed: %d" % rc)
    return out, st


# ---- scenes in the reference's dialect (sdfr_load_scene_hlsl), built for the CPU -----------------------------------------
_HLSL_DIR = os.path.join(_HERE, "_hlsl")


def build_hlsl(name, hlsl_text):
    """Translates `hlsl_text` with the library (sdfr_translate_scene_hlsl), compiles it with g++ around the product's per-pixel
    pipeline (hlsl_host.cpp) and returns the loaded library.  The VAR_ macros are the ones the run-time compiler would
    generate: slot k = order of first appearance."""
    import re
    import sys

    root = os.path.dirname(os.path.dirname(_HERE))
    if root not in sys.path:
        sys.path.insert(0, root)
    import sdf_playground_amd as sp

    os.makedirs(_HLSL_DIR, exist_ok=True)
    gen = os.path.join(_HLSL_DIR, name + ".scene.inc")
    slots = []
    for m in re.finditer(r"VAR_(\w+)\s*\(", hlsl_text):
        if m.group(1) not in slots:
            slots.append(m.group(1))
    text = "".join("#define VAR_%s(...) (U.scene_var[%d])\n" % (n, k) for k, n in enumerate(slots)) + sp.translate_scene_hlsl(hlsl_text)
    if not os.path.exists(gen) or open(gen).read() != text:
        with open(gen, "w") as f:
            f.write(text)
    so = os.path.join(_HLSL_DIR, "lib%s.so" % name)
    deps = [gen, os.path.join(_HERE, "hlsl_host.cpp")] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".h", ".inl"))]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-mfma", "-mavx2", "-fno-math-errno", "-Wno-unknown-pragmas", "-w",
                        "-pthread", "-I" + _CSRC, '-DSDFR_HLSL_SCENE_FILE="%s"' % gen, "-shared", "-o", so, os.path.join(_HERE, "hlsl_host.cpp")], check=True)
    L = ctypes.CDLL(so)
    assert L.hlslsim_frame_size() == ctypes.sizeof(FrameU)
    return L, slots


def render_hlsl(L, frame, stats=True, nthreads=None):
    W, H = frame.width, frame.height
    out = np.zeros((H, W, 4), np.float32)
    st = np.zeros((H, W, 3), np.uint32) if stats else None
    rc = L.hlslsim_render(ctypes.byref(frame), out.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p) if stats else None,
                          nthreads or os.cpu_count() or 1)
    assert rc == 0
    return out, st


def build_scene_source(name, cpp_text):
    """The same harness for a run-time scene in the library's C++ form (`struct Scene`, sdf_playground_amd/scenes/README.md)."""
    import re

    os.makedirs(_HLSL_DIR, exist_ok=True)
    gen = os.path.join(_HLSL_DIR, name + ".scene.inc")
    slots = []
    for m in re.finditer(r"VAR_(\w+)\s*\(", cpp_text):
        if m.group(1) not in slots:
            slots.append(m.group(1))
    text = "".join("#define VAR_%s(...) (U.scene_var[%d])\n" % (n, k) for k, n in enumerate(slots)) + cpp_text
    if not os.path.exists(gen) or open(gen).read() != text:
        with open(gen, "w") as f:
            f.write(text)
    so = os.path.join(_HLSL_DIR, "lib%s.so" % name)
    deps = [gen, os.path.join(_HERE, "hlsl_host.cpp")] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".h", ".inl"))]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-mfma", "-mavx2", "-fno-math-errno", "-Wno-unknown-pragmas", "-w",
                        "-pthread", "-I" + _CSRC, '-DSDFR_HLSL_SCENE_FILE="%s"' % gen, "-DSDFR_SCENE_HAS_PREPARE=1", "-shared", "-o", so,
                        os.path.join(_HERE, "hlsl_host.cpp")], check=True)
    L = ctypes.CDLL(so)
    assert L.hlslsim_frame_size() == ctypes.sizeof(FrameU)
    return L, slots


render_scene_source = render_hlsl
