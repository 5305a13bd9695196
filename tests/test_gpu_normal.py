"""GPU tier: the scene ABI's map_normal / NormalOutput (sdf_structs.hlsl:39-52; pshader_sdf.hlsl:318-330, :520) and the
driver's epsilons (pshader_sdf.hlsl:31-35) as run-time limits, on the HIP kernels through the C ABI against the oracle:
the diagnostic scene "normal_test" (Scene::normal: an analytic normal on one object, a wider normal_sample_dist on two
others) on both schedules, with the step shortcuts on and off, and its text compiled at run time; every built-in scene
with all five epsilons off their defaults; and what sdfr_set_limits accepts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_normal_cpu import CAMS, EPS_DEFAULT, EPS_OTHER, H, W, _frame


def _set_camera(sp, r, cam, w, h):
    c = sp.Camera()
    c.SetEye(cam[0])
    c.SetLookat(cam[1])
    c.SetAspect(float(np.float32(w) / np.float32(h)))
    r.setCamera(c)


def _check(r, oracle, scene, f, schedules, shortcuts=(False,)):
    import sdf_playground_amd as sp

    ref, rst, tot = oracle.render(scene, f, stats=True)
    for schedule in schedules:
        r.setSchedule(schedule)
        for sc in shortcuts:
            r.setStepShortcuts(sc)
            img, st = r.render(None, f.width, f.height, pixel_stats=True)
            same = np.array_equal(img.view(np.uint32), ref.view(np.uint32)) or np.array_equal(img, ref, equal_nan=True)
            assert same, (scene, schedule, sc, int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()))
            assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]), (scene, schedule, sc)
            if sc and schedule == sp.SCHEDULE_PIXEL:
                assert (st[..., 1] <= rst[..., 1]).all()
            else:
                assert np.array_equal(st[..., 1], rst[..., 1]), (scene, schedule, sc)
                s = r.getStats()
                assert (s.pixels, s.rays, s.march_evals, s.hits) == tuple(int(x) for x in tot)
    r.setStepShortcuts(False)
    r.setSchedule(sp.SCHEDULE_PIXEL)


VARIANTS = [{}, dict(round=0.04), dict(analytic=0.0, round=0.0001), dict(max_cost_default=9, extension_lights=7), dict(EPS_OTHER), dict(debug_ny=1.0, debug_y=0.45)]


def _apply(r, oracle, scene, extra):
    names = {row[0] for row in oracle.var_table(scene)}
    r.resetVariables()
    lim = dict(iter_count=100, bounce_count=16, ray_count=8, light_count=8, range=100.0, max_cost_default=7, extension_lights=0,
               extension_marble_reflection=0.0, **EPS_DEFAULT)
    for k, v in extra.items():
        if k in names:
            assert r.setValue(k, v)
        else:
            lim[k] = v
    r.setLimits(**lim)


@pytest.mark.parametrize("cam", range(len(CAMS)))
@pytest.mark.parametrize("variant", range(len(VARIANTS)))
def test_normal_test_both_schedules(oracle, cam, variant):
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    try:
        r.initShader("normal_test")
        assert r.currentScene() == "normal_test" and "normal_test" not in sp.scene_names()
        assert [v.name for v in r.getVariableMap().values() if not v.name.startswith(("debug_", "show_"))] == ["analytic", "round"]
        extra = VARIANTS[variant]
        _apply(r, oracle, "normal_test", extra)
        r.setParameters(0.4)
        _set_camera(sp, r, CAMS[cam], W, H)
        _check(r, oracle, "normal_test", _frame(oracle, "normal_test", CAMS[cam], **extra), (sp.SCHEDULE_PIXEL, sp.SCHEDULE_WAVEFRONT), (False, True))
    finally:
        r.close()


def test_normal_test_compiled_at_run_time(oracle):
    """the same text through hiprtc: Scene::normal is found by the run-time kernel's SceneNormal trait as well"""
    import jit_util
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    try:
        text = jit_util.aot_scene_source("SceneNormalTest", files=("sdfr_scene_debug.h",))
        assert "static SDF_HD void normal(" in text
        r.initShaderSource("normal_test_rt", text)
        r.setParameters(0.4)
        for cam, extra in ((0, {}), (1, dict(round=0.04)), (2, dict(analytic=0.0)), (3, dict(EPS_OTHER)), (1, dict(debug_ny=1.0, debug_y=0.45))):
            _apply(r, oracle, "normal_test", extra)
            _set_camera(sp, r, CAMS[cam], W, H)
            _check(r, oracle, "normal_test", _frame(oracle, "normal_test", CAMS[cam], **extra), (sp.SCHEDULE_PIXEL,), (False, True))
    finally:
        r.close()


def test_every_scene_at_other_epsilons(oracle):
    """all five epsilons off their defaults, every scene, both schedules, shortcuts on and off"""
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    rng = np.random.default_rng(78)
    try:
        for scene in sp.scene_names() + ["debug_materials", "normal_test"]:
            r.initShader(scene)
            _apply(r, oracle, scene, EPS_OTHER)
            r.setParameters(0.8)
            eye = (float(rng.uniform(-5, 5)), float(rng.uniform(0.4, 5)), float(rng.uniform(-6, -2)))
            cam = (eye, (0.0, 1.0, 0.0))
            w, h = 96, 64
            _set_camera(sp, r, cam, w, h)
            _check(r, oracle, scene, _frame(oracle, scene, cam, stime=0.8, w=w, h=h, **EPS_OTHER), (sp.SCHEDULE_PIXEL, sp.SCHEDULE_WAVEFRONT), (False, True))
    finally:
        r.close()


def test_limits_carry_the_reference_epsilons_and_refuse_nonsense():
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    try:
        l = r.getLimits()
        for k, v in EPS_DEFAULT.items():
            assert getattr(l, k) == np.float32(v), k  # pshader_sdf.hlsl:31-35
        for bad in (dict(dist_eps=0.0), dict(dist_eps=0.01), dict(dist_eps=float("nan")), dict(grad_eps=0.0), dict(grad_eps=-1e-4), dict(reflect_eps=-1e-3),
                    dict(refract_eps=2.0), dict(shadow_eps=float("nan"))):
            with pytest.raises(sp.SdfrError):
                r.setLimits(**bad)
        for k, v in EPS_DEFAULT.items():
            assert getattr(r.getLimits(), k) == np.float32(v), k  # a refused call changes nothing
        r.setLimits(**EPS_OTHER)
        for k, v in EPS_OTHER.items():
            assert getattr(r.getLimits(), k) == np.float32(v), k
    finally:
        r.close()
