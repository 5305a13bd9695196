"""CPU tier: step shortcuts (FrameU::step_shortcuts, the library's default) against every step marched, on the host build
of the product's pipeline (tests/hostsim) -- for EVERY scene, from cameras the fixed views of the GPU tier do not take:
under the floor, exactly on it, grazing it, inside the scenes' bounding balls, far away, looking up and down.  Pixels,
ray counts and hit counts must be identical; only the step counters may fall short.

Found by review in round 2: Scene::ray_escapes rules built on ray_leaves_floor_and_ball (sdfr_lib.h) took the floor to be
behind every ray that does not descend -- but from p.y <= 0 the reference's fast plane, p.y / 1e-20, is <= 0 and the ray
HITS the floor at its first sample (sdf_primitives.hlsl:59-70, pshader_sdf.hlsl:207-214)."""
import numpy as np
import pytest

FOVY = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
W, H = 48, 32


def _views(rng):
    views = [
        ((3.0, -1.0, -5.0), (0.0, 1.0, 0.0)),      # under the floor, looking up through it
        ((0.5, -0.25, 0.3), (4.0, 3.0, 1.0)),
        ((2.0, 0.0, -3.0), (0.0, 1.0, 0.0)),       # exactly on the floor
        ((2.0, 0.0, -3.0), (5.0, 0.0, 4.0)),       # ... looking along it
        ((-4.0, 1e-22, 1.0), (0.0, 0.6, 0.0)),     # a hair above: the fast plane is 0.01 there
        ((3.0, 0.02, 3.0), (0.0, 0.05, 0.0)),      # grazing
        ((0.1, 1.0, 0.05), (3.0, 1.5, 2.0)),       # inside the bounding balls about (0, 1, 0)
        ((0.0, 1.9, 0.2), (0.0, 9.0, 0.0)),        # straight up from inside
        ((40.0, 25.0, -30.0), (0.0, 1.0, 0.0)),    # far away
        ((1.0, 9.0, 1.0), (1.2, 0.0, 1.1)),        # straight down
    ]
    for _ in range(3):
        eye = (float(rng.uniform(-9, 9)), float(rng.uniform(-1.5, 6)), float(rng.uniform(-9, 9)))
        tgt = (float(rng.uniform(-2, 2)), float(rng.uniform(-1, 4)), float(rng.uniform(-2, 2)))
        views.append((eye, tgt))
    return views


def _scene_names():
    import hostsim
    import ctypes

    L = hostsim.lib()
    L.hostsim_scene_count.restype = ctypes.c_int
    L.hostsim_scene_name.restype = ctypes.c_char_p
    L.hostsim_scene_name.argtypes = [ctypes.c_int]
    return [L.hostsim_scene_name(i).decode() for i in range(L.hostsim_scene_count())]


def test_every_scene_keeps_pixels_rays_and_hits_with_shortcuts(oracle):
    import hostsim

    rng = np.random.default_rng(20261004)
    saved = {}
    for scene in _scene_names():
        heavy = scene in ("tree", "terrain", "distortion", "tiling")
        for k, (eye, at) in enumerate(_views(rng)):
            if heavy and k % 3 != 0:
                continue
            f = oracle.default_frame(scene, W, H, basis=oracle.camera_lookat(eye, at, FOVY, np.float32(W) / np.float32(H)), stime=0.37 * k)
            if k % 4 == 1:
                f.max_cost_default, f.extension_lights = 9, 7
            hf = hostsim.frame_from_oracle(f)
            hf.step_shortcuts = 0
            img0, st0 = hostsim.render(scene, hf)
            hf.step_shortcuts = 1
            img1, st1 = hostsim.render(scene, hf)
            same = np.array_equal(img0.view(np.uint32), img1.view(np.uint32)) or np.array_equal(img0, img1, equal_nan=True)
            assert same, (scene, eye, at, int((img0.view(np.uint32) != img1.view(np.uint32)).any(axis=2).sum()))
            assert np.array_equal(st0[..., 0], st1[..., 0]) and np.array_equal(st0[..., 2], st1[..., 2]), (scene, eye, at)
            assert (st1[..., 1] <= st0[..., 1]).all(), (scene, eye, at)
            saved[scene] = saved.get(scene, 0) + int(st0[..., 1].sum()) - int(st1[..., 1].sum())
    # the rules do fire on these views (a test that compares two identical code paths proves nothing)
    for scene in ("fast_sphere", "cube_sea", "labyrinth", "fractal", "gems", "cube", "sierpinski", "table", "light_shadows", "tiling", "terrain", "gyroid", "fractal2", "neon",
                  "spiral", "shell", "basic_clouds"):
        assert saved[scene] > 0, scene


@pytest.mark.parametrize("scene", ["fast_sphere", "fractal", "gems", "cube"])
def test_camera_under_the_floor_matches_the_oracle_with_shortcuts(oracle, scene):
    """the case of the round-2 review, against the oracle itself: eye (3, -1, -5), every primary ray starts under the floor"""
    import hostsim

    f = oracle.default_frame(scene, 64, 48, basis=oracle.camera_lookat((3.0, -1.0, -5.0), (0.0, 1.0, 0.0), FOVY, np.float32(64.0 / 48.0)), stime=0.5)
    ref, rst, _ = oracle.render(scene, f, stats=True)
    assert int(rst[..., 2].sum()) >= 64 * 48  # the reference hits the floor from below in every pixel
    hf = hostsim.frame_from_oracle(f)
    hf.step_shortcuts = 1
    img, st = hostsim.render(scene, hf)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2])


@pytest.mark.parametrize("bounces,slots", [(16, 8), (9, 8), (5, 8), (3, 8), (2, 8), (1, 8), (16, 4), (16, 1), (6, 3)])
def test_shadow_rays_delivered_from_the_light_loop_keep_budget_and_order(oracle, bounces, slots):
    """gems declares inline_escaped_shadows (sdfr_pixel.h): a floor pixel's escaped shadow rays never enter the queue.  With eight
    lights, a ray budget that ends among them and a queue shorter than their number, pixels, ray and hit counts stay the oracle's."""
    import hostsim

    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    for eye, at in [((2.5, 2.0, 0.5), (0.0, 1.0, 0.0)), ((0.3, 0.4, -4.0), (0.0, 0.8, 0.0)), ((5.0, 6.0, 5.0), (0.0, 0.0, 0.0))]:
        f = oracle.default_frame("gems", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=1.3)
        f.max_cost_default, f.extension_lights = 9, 7
        f.bounce_count, f.ray_count = bounces, slots
        ref, rst, _ = oracle.render("gems", f, stats=True)
        hf = hostsim.frame_from_oracle(f)
        hf.step_shortcuts = 1
        img, st = hostsim.render("gems", hf)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, bounces, slots)
        assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]), (eye, bounces, slots)
        assert bounces < 3 or (st[..., 1] < rst[..., 1]).any()  # the rule fired: fewer evaluations than the reference makes


@pytest.mark.parametrize("ball", [None, (3.0, 1.5, 4.0), (-4.0, -3.0, 0.5), (0.0, 0.0, 25.0)])
def test_lense_shadow_rays_end_between_the_blob_fields(oracle, ball):
    """lense declares escapes_from (sdfr_pixel.h): a shadow ray towards an extension light (3 high, between the blob fields) is a miss
    once it is inside the free slab and has lens, light ball and pane behind it or aside.  Pixels, rays and hits stay the oracle's
    wherever the light ball is moved."""
    import hostsim

    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    fired = False
    for eye, at in [((0.0, 0.5, 7.0), (0.0, 0.0, 0.0)), ((6.0, -2.0, -6.0), (0.0, -4.0, 0.0)), ((1.0, 3.0, 1.0), (0.0, 5.0, -5.0)), ((0.2, 0.1, 0.3), (0.0, 0.0, -5.0))]:
        f = oracle.default_frame("lense", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=2.1)
        f.max_cost_default, f.extension_lights, f.bounce_count = 9, 7, 16
        if ball is not None:
            f.scene_var[0], f.scene_var[1], f.scene_var[2] = ball
        ref, rst, _ = oracle.render("lense", f, stats=True)
        hf = hostsim.frame_from_oracle(f)
        hf.step_shortcuts = 1
        img, st = hostsim.render("lense", hf)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, ball)
        assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]), (eye, ball)
        fired = fired or bool((st[..., 1] < rst[..., 1]).any())
    assert fired
