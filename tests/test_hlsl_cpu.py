"""CPU tier: scenes in the REFERENCE'S OWN DIALECT (sdfr_load_scene_hlsl; csrc/sdfr_hlsl.h, sdfr_hlsl.cpp).

The plugin the reference exposes for the raymarch path is an .hlsl scene file (pshader_sdf.hlsl:79-84; Application.cpp:229,320).
* every one of the 22 files under /root/reference/Engine/shader/scenes -- read here, in the build container, at test time; they
  are never copied into this repository and the tests skip where the directory is absent -- translates, compiles for gfx950
  (hiprtc, no device needed) and, built for the CPU around the product's per-pixel pipeline (tests/hostsim), renders the same
  bits as the oracle's restatement of that scene: pixels and ray / step / hit counters, from two cameras, with the scene's
  variables moved.  That is also a pin of the oracle's 22 scene restatements by the reference's own scene text;
* the textual pass, rule by rule, on builder-written snippets;
* two builder-written scenes in the dialect (sdf_playground_amd/scenes/pendulum.hlsl, rounded.hlsl) against their C++ twins."""
import glob
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

REF_SCENES = "/root/reference/Engine/shader/scenes"
HERE = os.path.dirname(os.path.abspath(__file__))
SCENES_DIR = os.path.join(os.path.dirname(HERE), "sdf_playground_amd", "scenes")
FOVY = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
needs_reference = pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="the reference tree is not here (GPU box): its scene files are read, never copied")


def _ref_names():
    return [os.path.basename(f)[len("sdf_scene_"):-len(".hlsl")] for f in sorted(glob.glob(os.path.join(REF_SCENES, "sdf_scene_*.hlsl")))]


def _ref_text(name):
    with open(os.path.join(REF_SCENES, "sdf_scene_%s.hlsl" % name)) as f:
        return f.read()


# ---- the textual pass -------------------------------------------------------------------------------------------
def test_translation_rules():
    import sdf_playground_amd as sp

    src = """#include "sdf_primitives.hlsl"
static const float k = 0.65;
void helper(float3 p, out float a, inout float2 b, in float c) { a = 1e30 + .5 + 5. + 1.f + 2.0f + 3 + 0x10; }
void map_light(GeometryInput input, inout LightOutput output[LIGHT_COUNT], inout float ambient_lighting_factor) { [unroll] for (uint i = 0; i < 2; ++i) { int n = (int)(p.x * 4.5); } }
"""
    t = sp.translate_scene_hlsl(src)
    body = t[t.index('#line 1 "scene.hlsl"'):]
    assert "#include \"sdf_primitives.hlsl\"" not in body and "static" not in body and "[unroll]" not in body
    assert "const float k = 0.65f;" in body
    assert "void helper(float3 p, float &a, float2 &b, float c)" in body
    assert "a = 1e30f + .5f + 5.f + 1.f + 2.0f + 3 + 0x10;" in body           # floats get the suffix, integers and suffixed literals stay
    assert "LightOutput output[LIGHT_COUNT], float &ambient_lighting_factor" in body  # an array parameter is a reference already
    assert "int n = ftoi_(p.x * 4.5f);" in body                               # D3D's saturating float -> int cast
    assert "p.x" in body and "p.xf" not in body                                # swizzles are not numbers
    assert t.rstrip().endswith("typedef hlsl::SceneAdapter<hlsl::UserScene> Scene;")


def test_a_scene_that_does_not_compile_reports_the_compiler_s_words():
    import sdf_playground_amd as sp

    ok, log = sp.check_scene_hlsl("void map(GeometryInput geometry) { float3 p = geometry.pos.qq; }")
    assert not ok and "scene.hlsl" in log and "qq" in log


# ---- the reference's own scene files ---------------------------------------------------------------------------
@needs_reference
def test_all_22_reference_scene_files_are_there():
    assert len(_ref_names()) == 22


@needs_reference
def test_reference_scene_files_compile_for_gfx950():
    """sdfr_check_scene_hlsl on every file as it is (hiprtc; no device): ~1.5 s each"""
    import sdf_playground_amd as sp

    failed = {}
    for name in _ref_names():
        ok, log = sp.check_scene_hlsl(_ref_text(name))
        if not ok:
            failed[name] = [l for l in log.splitlines() if "error" in l][:4]
    assert not failed, failed


@pytest.fixture(scope="module")
def reference_scene_libs():
    import hostsim

    names = _ref_names()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        libs = list(ex.map(lambda n: hostsim.build_hlsl("ref_" + n, _ref_text(n)), names))
    return dict(zip(names, libs))


@needs_reference
def test_reference_scene_files_render_the_oracle_s_bits(oracle, reference_scene_libs):
    import hostsim

    rng = np.random.default_rng(2026)
    for name, (L, slots) in reference_scene_libs.items():
        table = oracle.var_table(name)
        # the variable slots the generated class reads (order of first appearance in the text) are the oracle's
        assert slots == [r[0] for r in sorted(table, key=lambda r: r[6]) if r[6] >= 0], name
        heavy = name in ("tree", "terrain", "distortion", "tiling")
        w, h = (40, 28) if heavy else (72, 48)
        for k in range(2):
            if k == 0:
                f = oracle.default_frame(name, w, h, stime=0.5)
            else:
                eye = (float(rng.uniform(-5, 5)), float(rng.uniform(0.4, 5)), float(rng.uniform(-6, -2)))
                f = oracle.default_frame(name, w, h, basis=oracle.camera_lookat(eye, (0.0, 1.0, 0.0), FOVY, np.float32(w) / np.float32(h)), stime=1.7)
                for vname, mn, mx, _start, _step, _v, slot in table:
                    if slot >= 0:
                        f.scene_var[slot] = float(np.float32(rng.uniform(mn, mx)))
                f.max_cost_default = 9
                f.debug_ny, f.debug_y = (1.0, 0.6) if name == "cube" else (0.0, 0.0)  # once through the debug-plane build
            ref, rst, _ = oracle.render(name, f, stats=True)
            img, st = hostsim.render_hlsl(L, hostsim.frame_from_oracle(f))
            same = np.array_equal(img.view(np.uint32), ref.view(np.uint32)) or np.array_equal(img, ref, equal_nan=True)
            assert same, (name, k, int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()))
            assert np.array_equal(st, rst), (name, k)


# ---- builder-written scenes in the dialect -----------------------------------------------------------------------
def test_rounded_hlsl_is_normal_test(oracle):
    """map_normal in the dialect (use_normal, normal_sample_dist) = the library's normal_test scene = the oracle's"""
    import hostsim
    from test_normal_cpu import CAMS, _frame

    L, slots = hostsim.build_hlsl("rounded", open(os.path.join(SCENES_DIR, "rounded.hlsl")).read())
    assert slots == ["round", "analytic"]
    for cam in CAMS[:3]:
        for extra in ({}, dict(round=0.04), dict(analytic=0.0, max_cost_default=9, extension_lights=7)):
            f = _frame(oracle, "normal_test", cam, **extra)
            ref, rst, _ = oracle.render("normal_test", f, stats=True)
            img, st = hostsim.render_hlsl(L, hostsim.frame_from_oracle(f))
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(st, rst), (cam, extra)


def test_pendulum_hlsl_equals_its_cpp_twin():
    import hostsim

    LH, slots = hostsim.build_hlsl("pendulum_hlsl", open(os.path.join(SCENES_DIR, "pendulum.hlsl")).read())
    LC, slots_c = hostsim.build_scene_source("pendulum_cpp", open(os.path.join(SCENES_DIR, "pendulum.scene.h")).read())
    assert slots == slots_c == ["swing", "rod", "radius"]
    for stime, values in ((0.3, (0.7, 1.8, 0.45)), (2.1, (1.1, 2.2, 0.7))):
        f = hostsim.FrameU()
        hostsim.lib().hostsim_frame_defaults(__import__("ctypes").byref(f))
        f.width, f.height, f.stime = 96, 64, stime
        basis = [(0.0, 2.0, -6.0), (0.0, -0.1, 0.99), (0.8, 0.0, 0.0), (0.0, 0.55, 0.06)]
        for i in range(3):
            f.eye[i], f.front[i], f.right[i], f.top[i] = basis[0][i], basis[1][i], basis[2][i], basis[3][i]
        for k, v in enumerate(values):
            f.scene_var[k] = v
        a, sa = hostsim.render_hlsl(LH, f)
        b, sb = hostsim.render_scene_source(LC, f)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(sa, sb)
        assert (sa[..., 0] >= 2).sum() > 200  # the mirror ball and the shadow rays are in the picture


# cameras for the two dialect scenes with oracle twins: (eye, lookat)
TWIN_CAMS = [((0.0, 2.0, -5.0), (0.0, 1.0, 0.0)), ((4.5, 1.2, -3.5), (0.5, 0.8, 0.5)), ((-3.0, 4.0, 4.5), (0.0, 0.6, 1.0)), ((0.3, 0.4, -14.0), (0.0, 1.0, 0.0))]


def twin_frame(oracle, scene, cam, w=96, h=64, stime=0.7, **extra):
    f = oracle.default_frame(scene, w, h, basis=oracle.camera_lookat(cam[0], cam[1], FOVY, np.float32(w) / np.float32(h)), stime=stime)
    slots = {r[0]: r[6] for r in oracle.var_table(scene)}
    for k, v in extra.items():
        if k in slots:
            if slots[k] >= 0:
                f.scene_var[slots[k]] = v
            else:
                setattr(f, k, v)
        else:
            setattr(f, k, v)
    return f


TWIN_CASES = {
    # 2-D / 4-D simplex noise, grad4; camera_distance and the ray offsets read in the geometry step (pshader_sdf.hlsl:187-218)
    "noise_lod": [dict(), dict(lod=3.0, freq=6.5, bump=0.04), dict(lod=40.0, stime=3.9, max_cost_default=9), dict(debug_ny=1.0, debug_y=0.8)],
    # swizzled l-values, inout swizzles, float3x3 + mul, static const initialisers, saturating (int), a voronoi overload, VAR_ quirks
    "dialect_tour": [dict(), dict(spin=-1.3, reach=2.7, blend=0.25, shine=0.8), dict(stime=5.2, extension_lights=5, max_cost_default=9), dict(debug_nx=1.0, debug_x=0.3)],
}


@pytest.mark.parametrize("scene", sorted(TWIN_CASES))
def test_dialect_scene_renders_its_oracle_twin_s_bits(oracle, scene):
    import hostsim

    text = open(os.path.join(SCENES_DIR, scene + ".hlsl")).read()
    L, slots = hostsim.build_hlsl(scene, text)
    table = oracle.var_table(scene)
    assert slots == [r[0] for r in sorted(table, key=lambda r: r[6]) if r[6] >= 0]
    # the table the library parses from the text is the one the oracle parses from the same tags (ShaderUtil.cpp:122-191)
    parsed = oracle.parse_vars(text)
    for name, mn, mx, start, step, _v, slot in table:
        if slot >= 0:
            assert parsed[name][:4] == (mn, mx, start, step), name
    hits = 0
    for cam, extra in zip(TWIN_CAMS, TWIN_CASES[scene]):
        f = twin_frame(oracle, scene, cam, **extra)
        ref, rst, _ = oracle.render(scene, f, stats=True)
        img, st = hostsim.render_hlsl(L, hostsim.frame_from_oracle(f))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (scene, cam, extra, int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()))
        assert np.array_equal(st, rst), (scene, cam, extra)
        hits += int(rst[..., 2].sum())
    assert hits > 5000


def test_builder_written_dialect_scenes_compile_for_gfx950():
    import sdf_playground_amd as sp

    for name in ("pendulum", "rounded", "noise_lod", "dialect_tour"):
        ok, log = sp.check_scene_hlsl(open(os.path.join(SCENES_DIR, name + ".hlsl")).read())
        assert ok, (name, log[-2000:])


def test_geometry_step_sees_the_march_state(oracle):
    """noise_lod's picture depends on geometry.camera_distance and the ray offsets DURING the march: with `lod` below the ball's
    distance the bump is gone, and the ring is thicker than its 1-cm tube wherever a pixel is wider than that"""
    f_near = twin_frame(oracle, "noise_lod", TWIN_CAMS[0], lod=40.0)
    f_far = twin_frame(oracle, "noise_lod", TWIN_CAMS[0], lod=0.0)
    a, _, _ = oracle.render("noise_lod", f_near)
    b, _, _ = oracle.render("noise_lod", f_far)
    assert (np.abs(a - b).max(axis=2) > 1e-3).mean() > 0.02
    # far away the ring would fall between the pixels without the footprint term; with it the ring still collects hits
    f = twin_frame(oracle, "noise_lod", ((-2.4, 1.0, -30.0), (-2.4, 1.0, 0.3)), w=120, h=80)
    _, st, _ = oracle.render("noise_lod", f, stats=True)
    ring_rows = st[30:50, 40:80, 2]
    assert ring_rows.sum() > 20


def test_bare_integer_casts_are_translated():
    import sdf_playground_amd as sp

    t = sp.translate_scene_hlsl("void map_normal(GeometryInput g, inout NormalOutput o) { int a = (int)cell.x; uint b = (uint) index; int c = (int)f(x); int d = (int)(y); int e = (int)v[2]; }")
    assert "int a = ftoi_(cell.x);" in t and "uint b = ftou_(index);" in t and "int d = ftoi_(y);" in t
    assert "(int)f(x)" in t and "(int)v[2]" in t  # calls and elements keep the C cast
