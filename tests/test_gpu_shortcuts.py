"""GPU tier: step shortcuts (sdfr_set_step_shortcuts, the library's default): a ray its scene knows to be a miss already
stops marching.  Pixels, ray counts and hit counts must not move by a bit; only the step counters fall short of the
oracle's.  The rest of the suite runs with the shortcuts off (tests/conftest.py) so that it can compare step counters."""
import numpy as np
import pytest

from test_gpu_parity import _setup

pytestmark = pytest.mark.gpu

VIEWS = [  # eye, look-at: above the cubes looking over them, from the floor up, straight down, along the slab, far away
    ((3.0, 4.5, 1.0), (6.0, 3.2, 4.0)), ((0.3, 0.4, 0.2), (2.0, 3.0, 1.5)), ((1.0, 9.0, 1.0), (1.2, 0.0, 1.1)),
    ((0.0, 2.0, 0.0), (8.0, 2.1, 3.0)), ((30.0, 6.0, -25.0), (0.0, 2.0, 0.0)), ((2.0, 3.7, 2.0), (9.0, 3.9, 2.5)),
]


@pytest.fixture(scope="module")
def renderer():
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    yield r
    r.close()


def _render(r, oracle, scene, eye, at, w, h, stime, limits, shortcuts):
    import torch
    import sdf_playground_amd as sp

    fovy = np.float32(sp.to_radian(60.0))
    f = oracle.default_frame(scene, w, h, basis=oracle.camera_lookat(eye, at, fovy, np.float32(w) / np.float32(h)), stime=stime)
    for k, v in limits.items():
        setattr(f, k, v)
    r.setLimits(**limits)
    r.setParameters(stime)
    cam = sp.Camera()
    cam.SetEye(eye)
    cam.SetLookat(at)
    cam.SetFOVY(float(fovy))
    cam.SetAspect(w / h)
    r.setStepShortcuts(shortcuts)
    img, st = r.render(cam, w, h, pixel_stats=True)
    return f, img, st, r.getStats()


@pytest.mark.parametrize("view", range(len(VIEWS)))
def test_cube_sea_shortcuts_change_no_pixel(renderer, oracle, view):
    eye, at = VIEWS[view]
    renderer.initShader("cube_sea")
    limits = dict(iter_count=128, max_cost_default=6)
    try:
        f, img, st, tot = _render(renderer, oracle, "cube_sea", eye, at, 160, 96, 0.35 * view, limits, True)
        ref, rst, _ = oracle.render("cube_sea", f, stats=True)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2])  # rays, hits
        assert (st[..., 1] <= rst[..., 1]).all()
        assert tot.march_evals == int(st[..., 1].sum(dtype=np.int64))
        # and off: the counters are the oracle's
        _, img0, st0, _ = _render(renderer, oracle, "cube_sea", eye, at, 160, 96, 0.35 * view, limits, False)
        assert np.array_equal(img0.view(np.uint32), ref.view(np.uint32)) and np.array_equal(st0, rst)
        if view in (0, 1, 3, 5):
            assert int(st[..., 1].sum()) < int(rst[..., 1].sum())  # something was saved where rays leave upwards
    finally:
        renderer.setStepShortcuts(False)


@pytest.mark.parametrize("scene,limits", [("labyrinth", dict(iter_count=256)), ("labyrinth", dict(extension_marble_reflection=0.25)),
                                          ("fractal", dict(iter_count=512)), ("gems", dict(max_cost_default=9, extension_lights=7)),
                                          ("tree", None), ("terrain", None), ("distortion", None), ("fast_sphere", None), ("cube", None), ("sierpinski", None),
                                          ("basic_transparency", None), ("coordinate_material", None), ("table", None),
                                          ("light_shadows", None), ("tiling", None), ("gyroid", None), ("fractal2", None), ("neon", None), ("spiral", None), ("shell", None), ("basic_clouds", None),
                                          ("cube_sea", dict(max_cost_default=6))])
def test_scenes_with_an_escape_rule(renderer, oracle, scene, limits):
    """every scene that declares ray_escapes(), at the parity tests' camera and from six more (looking up from the
    floor, down from above, along the horizon, from under / on / a hair above the floor): shortcuts on changes no pixel, ray or hit count and adds no step"""
    import sdf_playground_amd as sp

    try:
        f = _setup(renderer, oracle, scene, 0.75, limits=limits)
        # ... and from under the floor, exactly on it and a hair above it (round-2 review: rules that take the floor to be
        # behind a rising ray must know that the ray is above it -- tests/test_shortcuts_cpu.py)
        views = [None, ((0.4, 0.3, 0.3), (3.0, 4.0, 2.0)), ((1.0, 12.0, 1.0), (1.3, 0.0, 1.2)), ((-6.0, 2.3, -5.0), (6.0, 2.5, 7.0)),
                 ((3.0, -1.0, -5.0), (0.0, 1.0, 0.0)), ((2.0, 0.0, -3.0), (0.0, 1.0, 0.0)), ((-4.0, 1e-22, 1.0), (0.0, 0.6, 0.0))]
        fovy = np.float32(sp.to_radian(60.0))
        for view in views:
            if view is not None:
                eye, at = view
                basis = oracle.camera_lookat(eye, at, fovy, np.float32(f.width) / np.float32(f.height))
                for name, row in zip(("eye", "front", "right", "top"), basis):
                    getattr(f, name)[:] = [float(x) for x in row]
                cam = sp.Camera()
                cam.SetEye(eye)
                cam.SetLookat(at)
                cam.SetFOVY(float(fovy))
                cam.SetAspect(f.width / f.height)
                renderer.setCamera(cam)
            ref, rst, _ = oracle.render(scene, f, stats=True)
            renderer.setStepShortcuts(True)
            img, st = renderer.render(None, f.width, f.height, pixel_stats=True)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (scene, view)
            assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]) and (st[..., 1] <= rst[..., 1]).all(), (scene, view)
            renderer.setStepShortcuts(False)
            img, st = renderer.render(None, f.width, f.height, pixel_stats=True)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(st, rst), (scene, view)
    finally:
        renderer.setStepShortcuts(False)


@pytest.mark.parametrize("config,scene,saves", [("2", "cube_sea", 0.7), ("3", "labyrinth", 0.9), ("4", "fractal", 0.9), ("5g", "gems", 0.7), ("3r", "labyrinth", 0.9)])
def test_configurations_at_full_size_with_shortcuts(renderer, oracle, config, scene, saves):
    """the BASELINE configurations whose scene has an escape rule, as bench.py runs them (shortcuts on): every 24th pixel
    against the oracle, rays and hits too; the whole frame equals the frame with every step marched; steps are saved"""
    import torch
    import bench
    import sdf_playground_amd as sp

    cfg = bench.CONFIGS[config]
    w, h, stride = cfg["width"], cfg["height"], 24
    renderer.initShader(scene)
    renderer.setLimits(iter_count=100, bounce_count=16, ray_count=8, light_count=8, range=100.0, max_cost_default=7, extension_lights=0,
                       extension_marble_reflection=0.0)
    renderer.setLimits(**cfg["limits"])
    cam, stime = bench.make_camera(5, w, h, config)
    renderer.setParameters(stime)
    f = bench.oracle_frame(oracle, 5, w, h, config)
    try:
        renderer.setStepShortcuts(True)
        img = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        pst = torch.empty((h, w, 3), dtype=torch.int32, device="cuda")
        renderer.render(cam, w, h, out=img, pixel_stats=pst)
        on = renderer.getStats()
        renderer.setStepShortcuts(False)
        img0 = torch.empty_like(img)
        renderer.render(cam, w, h, out=img0)
        off = renderer.getStats()
        assert torch.equal(img.view(torch.int32), img0.view(torch.int32))
        assert (on.pixels, on.rays, on.hits) == (off.pixels, off.rays, off.hits) and on.march_evals < saves * off.march_evals
        ref, rst, _ = oracle.render(scene, f, step=(stride, stride), stats=True)
        assert np.array_equal(img[::stride, ::stride].cpu().numpy().view(np.uint32), ref[::stride, ::stride].view(np.uint32))
        got = pst[::stride, ::stride].cpu().numpy()
        assert np.array_equal(got[..., 0], rst[::stride, ::stride, 0]) and np.array_equal(got[..., 2], rst[::stride, ::stride, 2])
    finally:
        renderer.setStepShortcuts(False)
        renderer.setLimits(iter_count=100, bounce_count=16, ray_count=8, light_count=8, range=100.0, max_cost_default=7, extension_lights=0,
                           extension_marble_reflection=0.0)


def test_debug_plane_and_wavefront_march_every_step(renderer, oracle):
    """the debug plane is an extra object (no shortcut in that build) and the wavefront schedule has no shortcuts:
    with them requested, both still give the oracle's counters"""
    import sdf_playground_amd as sp

    try:
        f = _setup(renderer, oracle, "cube_sea", 0.75, variables=dict(debug_ny=1.0, debug_y=4.5))
        renderer.setStepShortcuts(True)
        for schedule in (1, 0):
            renderer.setSchedule(schedule)
            img, st = renderer.render(None, f.width, f.height, pixel_stats=True)
            ref, rst, _ = oracle.render("cube_sea", f, stats=True)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(st, rst), schedule
        f = _setup(renderer, oracle, "cube_sea", 0.75)
        renderer.setStepShortcuts(True)
        renderer.setSchedule(0)
        img, st = renderer.render(None, f.width, f.height, pixel_stats=True)
        ref, rst, _ = oracle.render("cube_sea", f, stats=True)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(st, rst)
    finally:
        renderer.setSchedule(1)
        renderer.setStepShortcuts(False)


def test_run_time_scene_with_an_escape_rule(renderer, oracle):
    """cube_sea's text compiled at run time carries its ray_escapes() along"""
    from jit_util import aot_scene_source

    f = _setup(renderer, oracle, "cube_sea", 0.75)
    try:
        renderer.initShaderSource("cube_sea_rt", "// SDFR_FAST_EXACT_MATH\n" + aot_scene_source("SceneCubeSea"))
        ref, rst, _ = oracle.render("cube_sea", f, stats=True)
        for on in (True, False):
            renderer.setStepShortcuts(on)
            img, st = renderer.render(None, f.width, f.height, pixel_stats=True)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), on
            assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2])
            assert np.array_equal(st[..., 1], rst[..., 1]) if not on else (st[..., 1] <= rst[..., 1]).all()
    finally:
        renderer.setStepShortcuts(False)


@pytest.mark.parametrize("bounces,slots", [(16, 8), (9, 8), (5, 8), (3, 8), (2, 8), (1, 8), (16, 4), (16, 1), (6, 3)])
def test_delivered_shadow_rays_keep_budget_and_queue_length(renderer, oracle, bounces, slots):
    """gems declares inline_escaped_shadows: with eight lights, a ray budget that ends among a floor pixel's shadow rays and a queue
    shorter than their number, the HIP path renders the oracle's pixels, ray and hit counts (the CPU tier runs the same cases on
    the host build: tests/test_shortcuts_cpu.py)."""
    renderer.initShader("gems")
    limits = dict(max_cost_default=9, extension_lights=7, bounce_count=bounces, ray_count=slots)
    try:
        for k, (eye, at) in enumerate([((2.5, 2.0, 0.5), (0.0, 1.0, 0.0)), ((0.3, 0.4, -4.0), (0.0, 0.8, 0.0)), ((5.0, 6.0, 5.0), (0.0, 0.0, 0.0))]):
            f, img, st, _tot = _render(renderer, oracle, "gems", eye, at, 160, 96, 1.3 + k, limits, True)
            ref, rst, _ = oracle.render("gems", f, stats=True)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, bounces, slots)
            assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2])
            assert (st[..., 1] <= rst[..., 1]).all()
    finally:
        renderer.setStepShortcuts(False)
        renderer.setLimits(max_cost_default=7, extension_lights=0, bounce_count=16, ray_count=8)


@pytest.mark.parametrize("ball", [None, (3.0, 1.5, 4.0), (-4.0, -3.0, 0.5), (0.0, 0.0, 25.0)])
def test_lense_shadow_rays_end_between_the_blob_fields(renderer, oracle, ball):
    """lense declares escapes_from: shadow rays towards the extension lights end between the blob fields, wherever the light ball is"""
    renderer.initShader("lense")
    limits = dict(max_cost_default=9, extension_lights=7)
    saved = 0
    try:
        renderer.resetVariables()
        if ball is not None:
            for name, v in zip(("xpos", "ypos", "zpos"), ball):
                renderer.setValue(name, v)
        for eye, at in [((0.0, 0.5, 7.0), (0.0, 0.0, 0.0)), ((6.0, -2.0, -6.0), (0.0, -4.0, 0.0)), ((1.0, 3.0, 1.0), (0.0, 5.0, -5.0))]:
            fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
            f, img, st, _tot = _render(renderer, oracle, "lense", eye, at, 160, 96, 2.1, limits, True)
            if ball is not None:
                f.scene_var[0], f.scene_var[1], f.scene_var[2] = ball
            ref, rst, _ = oracle.render("lense", f, stats=True)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, ball)
            assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2])
            assert (st[..., 1] <= rst[..., 1]).all()
            saved += int(rst[..., 1].sum()) - int(st[..., 1].sum())
        assert saved > 0
    finally:
        renderer.setStepShortcuts(False)
        renderer.resetVariables()
        renderer.setLimits(max_cost_default=7, extension_lights=0)
