"""GPU tier: scenes in the reference's own dialect through sdfr_load_scene_hlsl (hiprtc) on the HIP kernels -- a builder-written
scene in that dialect against its C++ twin, and map_normal in the dialect against the built-in normal_test scene and the
oracle.  (The 22 scene files of the reference go through the same translation on the CPU tier, where the reference tree is:
tests/test_hlsl_cpu.py; the tree does not travel to the GPU box.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdf_playground_amd", "scenes")
W, H = 200, 136


def test_pendulum_hlsl_equals_its_cpp_twin():
    import sdf_playground_amd as sp

    a, b = sp.SDFRenderer(0), sp.SDFRenderer(0)
    try:
        a.initShaderHlsl("pendulum_hlsl", os.path.join(SCENES_DIR, "pendulum.hlsl"))
        b.initShaderSource("pendulum_cpp", os.path.join(SCENES_DIR, "pendulum.scene.h"))
        assert a.currentScene() == "pendulum_hlsl"
        ta = [(v.name, v.minval, v.maxval, v.start, v.step) for v in a.getVariableMap().values()]
        assert ta == [(v.name, v.minval, v.maxval, v.start, v.step) for v in b.getVariableMap().values()]
        assert [t[0] for t in ta if not t[0].startswith(("debug_", "show_"))] == ["radius", "rod", "swing"]  # std::map order
        cam = sp.Camera()
        cam.SetEye((0.5, 2.0, -6.0))
        cam.SetLookat((0.0, 1.5, 0.0))
        cam.SetAspect(W / H)
        for stime, values, extra in ((0.3, {}, {}), (2.1, dict(swing=1.1, rod=2.2, radius=0.7), dict(max_cost_default=9, extension_lights=7)),
                                     (1.0, dict(debug_ny=1.0, debug_y=0.5), {})):
            imgs = []
            for r in (a, b):
                r.setParameters(stime)
                r.resetVariables()
                r.setLimits(iter_count=100, max_cost_default=7, extension_lights=0)
                r.setLimits(**extra)
                for k, v in values.items():
                    assert r.setValue(k, v)
                for shortcuts in (False, True):
                    r.setStepShortcuts(shortcuts)
                    img, st = r.render(cam, W, H, pixel_stats=True)
                    imgs.append((img, st, r.getStats()))
            ref = imgs[0]
            for img, st, tot in imgs[1:]:
                assert np.array_equal(img.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(st, ref[1])
                assert (tot.pixels, tot.rays, tot.march_evals, tot.hits) == (ref[2].pixels, ref[2].rays, ref[2].march_evals, ref[2].hits)
            assert (ref[1][..., 0] >= 2).sum() > 500
    finally:
        a.close()
        b.close()


def test_map_normal_in_the_dialect(oracle):
    """rounded.hlsl (use_normal, normal_sample_dist in HLSL) = the built-in normal_test = the oracle"""
    import sdf_playground_amd as sp
    from test_normal_cpu import CAMS, _frame

    r, aot = sp.SDFRenderer(0), sp.SDFRenderer(0)
    try:
        r.initShaderHlsl("rounded", os.path.join(SCENES_DIR, "rounded.hlsl"))
        aot.initShader("normal_test")
        for cam, extra in ((CAMS[0], {}), (CAMS[1], dict(round=0.04)), (CAMS[2], dict(analytic=0.0))):
            f = _frame(oracle, "normal_test", cam, w=W, h=H, **extra)
            ref, rst, _ = oracle.render("normal_test", f, stats=True)
            c = sp.Camera()
            c.SetEye(cam[0])
            c.SetLookat(cam[1])
            c.SetAspect(float(np.float32(W) / np.float32(H)))
            for h_ in (r, aot):
                h_.setParameters(0.4)
                h_.resetVariables()
                for k, v in extra.items():
                    assert h_.setValue(k, v)
                img, st = h_.render(c, W, H, pixel_stats=True)
                assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(st, rst), (cam, extra)
    finally:
        r.close()
        aot.close()


def test_a_broken_scene_leaves_the_loaded_one_in_place():
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    try:
        r.initShader("fast_sphere")
        with pytest.raises(sp.SdfrError) as e:
            r.initShaderHlsl("broken", "void map(GeometryInput geometry) { float3 p = geometry.pos.qq; }")
        assert e.value.code == -7 and "scene.hlsl" in str(e.value)
        assert r.currentScene() == "fast_sphere"  # SceneManager.cpp:118-127: the old shader stays
    finally:
        r.close()


@pytest.mark.parametrize("scene", ["noise_lod", "dialect_tour"])
def test_dialect_scene_on_the_gpu_renders_its_oracle_twin_s_bits(oracle, scene):
    """Builder-written scenes in the reference's dialect, compiled by hiprtc and run by the pixel kernel, against their oracle
    twins (oracle/test_scenes.h) -- HIP against the oracle, not HIP against HIP.  noise_lod: snoise(float2), snoise(float4),
    grad4, and a geometry step that reads geometry.camera_distance and the ray offsets (pshader_sdf.hlsl:187-218).
    dialect_tour: swizzled l-values and swaps, an inout swizzle, float3x3 + mul, static const initialisers, (int) with D3D's
    saturation, a scene-local voronoi overload, opRepLim / opShell / smin, and every VAR_ tag quirk."""
    import sdf_playground_amd as sp
    from test_hlsl_cpu import TWIN_CAMS, TWIN_CASES, twin_frame

    r = sp.SDFRenderer(0)
    try:
        r.initShaderHlsl(scene, os.path.join(SCENES_DIR, scene + ".hlsl"))
        table = {row[0]: row for row in oracle.var_table(scene)}
        got = {v.name: (v.minval, v.maxval, v.start, v.step) for v in r.getVariableMap().values()}
        assert list(got) == sorted(table)  # std::map order
        for name, row in table.items():
            assert got[name] == tuple(np.float32(x) for x in row[1:5]), name
        limit_keys = ("max_cost_default", "extension_lights")
        for cam, extra in zip(TWIN_CAMS, TWIN_CASES[scene]):
            f = twin_frame(oracle, scene, cam, w=W, h=H, **extra)
            ref, rst, _ = oracle.render(scene, f, stats=True)
            c = sp.Camera()
            c.SetEye(cam[0])
            c.SetLookat(cam[1])
            c.SetAspect(float(np.float32(W) / np.float32(H)))
            r.setParameters(extra.get("stime", 0.7))
            r.resetVariables()
            r.setLimits(iter_count=100, max_cost_default=7, extension_lights=0)
            r.setLimits(**{k: v for k, v in extra.items() if k in limit_keys})
            for k, v in extra.items():
                if k not in limit_keys and k != "stime":
                    assert r.setValue(k, v), k
            for shortcuts in (False, True):
                r.setStepShortcuts(shortcuts)
                img, st = r.render(c, W, H, pixel_stats=True)
                assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (scene, cam, extra, int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()))
                assert np.array_equal(st, rst), (scene, cam, extra)
    finally:
        r.close()
