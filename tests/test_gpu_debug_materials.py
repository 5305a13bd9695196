"""GPU tier, SURVEY.md 8(f)-4: MATERIAL_ITER / PLAIN / NORMAL1 / NORMAL2 on the HIP kernels against
the oracle -- the library's diagnostic scene "debug_materials" on both schedules, and the same scene
text compiled at run time (hiprtc) -- with iter_count != 100 so that
iter_count_to_color(iter, ITER_COUNT - 1) is exercised (pshader_sdf.hlsl:430-455)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_debug_materials_cpu import CAMS, H, W, _frame


def _render_and_compare(r, oracle, cam, iter_count, extra, schedules):
    import sdf_playground_amd as sp

    f = _frame(oracle, CAMS[cam], iter_count, **extra)
    ref, rst, tot = oracle.render("debug_materials", f, stats=True)
    r.setParameters(0.4)
    r.setLimits(iter_count=iter_count, max_cost_default=extra.get("max_cost_default", 7))
    for k in ("debug_ny", "debug_y"):
        r.setValue(k, extra.get(k, 0.0))
    c = sp.Camera()
    c.SetEye(CAMS[cam][0])
    c.SetLookat(CAMS[cam][1])
    c.SetAspect(float(np.float32(W) / np.float32(H)))
    r.setCamera(c)
    for schedule in schedules:
        r.setSchedule(schedule)
        img, st = r.render(None, W, H, pixel_stats=True)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (cam, iter_count, extra, schedule)
        assert np.array_equal(st, rst)
        s = r.getStats()
        assert (s.pixels, s.rays, s.march_evals, s.hits) == tuple(int(x) for x in tot)
    return ref


@pytest.mark.parametrize("cam", range(len(CAMS)))
@pytest.mark.parametrize("iter_count,extra", [(100, {}), (37, {}), (250, dict(max_cost_default=9)), (64, dict(debug_ny=1.0, debug_y=0.3))])
def test_debug_materials_both_schedules(oracle, cam, iter_count, extra):
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    r.initShader("debug_materials")
    assert r.currentScene() == "debug_materials"
    ref = _render_and_compare(r, oracle, cam, iter_count, extra, (sp.SCHEDULE_PIXEL, sp.SCHEDULE_WAVEFRONT))
    if not extra:
        # all four views are in the picture: untone-mapped pixels (ITER / NORMAL1 / NORMAL2 set alpha 0) and tone-mapped ones
        assert (ref[..., 3] == 0).sum() > 300 and (ref[..., 3] == 1).sum() > 300
    r.close()


def test_debug_materials_compiled_at_run_time(oracle):
    import jit_util
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    r.initShaderSource("debug_materials_rt", jit_util.aot_scene_source("SceneDebugMaterials", files=("sdfr_scene_debug.h",)))
    assert r.currentScene() == "debug_materials_rt"
    for cam, iter_count, extra in ((0, 37, {}), (1, 100, {}), (2, 64, dict(debug_ny=1.0, debug_y=0.3))):
        _render_and_compare(r, oracle, cam, iter_count, extra, (sp.SCHEDULE_PIXEL,))
    r.close()
