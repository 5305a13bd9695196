"""CPU tier: the lower bounds the labyrinth functor uses to cull its vase and torch
(sdf_playground_amd/csrc/sdfr_scenes.h, SceneLabyrinth::dist) hold on dense random samples,
and culling leaves every pixel bit-identical to the oracle from cameras close to those
objects."""
import ctypes
import math

import numpy as np


def test_labyrinth_bounds_hold_on_random_samples():
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_labyrinth_bounds.restype = ctypes.c_longlong
    L.hostsim_check_labyrinth_bounds.argtypes = [ctypes.c_longlong, ctypes.c_uint]
    assert L.hostsim_check_labyrinth_bounds(3000000, 7) == 0


def test_culling_is_invisible_near_vases_and_torches(oracle):
    import hostsim

    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    views = [((6.2, 1.6, 1.4), (7.0, 1.0, 3.0)), ((9.6, 2.4, 1.2), (9.0, 1.2, 3.0)), ((5.4, 3.2, 1.2), (5.2, 3.0, 3.0)),
             ((3.0, 0.8, 0.5), (9.0, 1.5, 3.0)), ((7.5, 6.0, 7.5), (7.5, 0.0, 3.0)), ((-12.8, 2.5, -17.2), (-11.0, 2.0, -17.0))]
    for eye, at in views:
        for stime in (0.0, 0.9):
            f = oracle.default_frame("labyrinth", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=stime)
            f.iter_count = 256
            ref, rst, _ = oracle.render("labyrinth", f, stats=True)
            img, st = hostsim.render("labyrinth", hostsim.frame_from_oracle(f))
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at)
            assert np.array_equal(st, rst)


def test_lense_bounds_hold_and_culling_is_invisible(oracle):
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_lense_bounds.restype = ctypes.c_longlong
    L.hostsim_check_lense_bounds.argtypes = [ctypes.c_longlong, ctypes.c_uint]
    assert L.hostsim_check_lense_bounds(3000000, 11) == 0
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    # cameras close to the mirror pane (at (0, 0, -5)), through it and from the far side
    for eye, at in [((2.5, 1.0, -2.5), (0.0, 0.0, -5.0)), ((-1.5, 2.5, -8.0), (0.0, 0.0, -5.0)), ((0.3, 0.2, -3.2), (0.0, 0.5, -5.0)),
                    ((6.0, 0.5, 4.0), (0.0, 0.0, -5.0))]:
        for stime in (0.0, 2.3):
            f = oracle.default_frame("lense", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=stime)
            ref, rst, _ = oracle.render("lense", f, stats=True)
            img, st = hostsim.render("lense", hostsim.frame_from_oracle(f))
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at, stime)
            assert np.array_equal(st, rst)


def test_fractal_bound_holds_and_culling_is_invisible(oracle):
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_fractal_bounds.restype = ctypes.c_longlong
    L.hostsim_check_fractal_bounds.argtypes = [ctypes.c_longlong, ctypes.c_uint]
    assert L.hostsim_check_fractal_bounds(4000000, 5) == 0
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    for eye, at in [((2.2, 1.6, 0.3), (0.0, 1.0, 0.0)), ((0.9, 1.9, 0.9), (0.0, 1.0, 0.0)), ((0.0, 3.5, 0.1), (0.0, 1.0, 0.0)), ((6.0, 0.4, 5.0), (0.0, 1.0, 0.0))]:
        f = oracle.default_frame("fractal", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=0.0)
        f.iter_count = 512
        ref, rst, _ = oracle.render("fractal", f, stats=True)
        img, st = hostsim.render("fractal", hostsim.frame_from_oracle(f))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at)
        assert np.array_equal(st, rst)


def test_distortion_bound_holds_and_culling_is_invisible(oracle):
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_distortion_bounds.restype = ctypes.c_longlong
    L.hostsim_check_distortion_bounds.argtypes = [ctypes.c_longlong, ctypes.c_uint]
    assert L.hostsim_check_distortion_bounds(3000000, 3) == 0
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    for eye, at in [((0.8, 1.8, -2.5), (0.0, 1.5, 0.0)), ((0.2, 1.5, -0.35), (0.0, 1.5, 0.0)), ((2.5, 0.3, 0.02), (0.0, 1.5, 0.0)), ((-3.0, 4.0, 3.0), (0.0, 1.0, 0.0))]:
        f = oracle.default_frame("distortion", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=0.4)
        ref, rst, _ = oracle.render("distortion", f, stats=True)
        img, st = hostsim.render("distortion", hostsim.frame_from_oracle(f))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at)
        assert np.array_equal(st, rst)


def test_cube_sea_bound_holds_and_culling_is_invisible(oracle):
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_cube_sea_bounds.restype = ctypes.c_longlong
    L.hostsim_check_cube_sea_bounds.argtypes = [ctypes.c_longlong, ctypes.c_uint]
    assert L.hostsim_check_cube_sea_bounds(3000000, 9) == 0
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    # from above the cubes (the bench camera's height), looking up from between them, grazing their tops, from inside
    # the slab, straight down and straight up (guard divisors near zero), and far above the field
    views = [((3.0, 4.5, 0.0), (4.0, 4.05, 0.6)), ((0.9, 0.2, 0.9), (3.0, 3.0, 5.0)), ((0.0, 3.7, 0.0), (30.0, 3.6, 7.0)), ((1.0, 2.0, 1.0), (9.0, 2.2, 4.0)),
             ((0.5, 9.0, 0.5), (0.5, 0.0, 0.5001)), ((0.3, 0.1, 0.2), (0.3, 50.0, 0.2001)), ((5.0, 900.0, 5.0), (40.0, 0.0, 30.0)), ((2.0, 1200.0, 2.0), (2.5, 0.0, 2.0))]
    for eye, at in views:
        for stime, limits in ((0.0, {}), (1.7, dict(iter_count=128, max_cost_default=6))):
            f = oracle.default_frame("cube_sea", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=stime)
            for k, v in limits.items():
                setattr(f, k, v)
            ref, rst, _ = oracle.render("cube_sea", f, stats=True)
            img, st = hostsim.render("cube_sea", hostsim.frame_from_oracle(f))
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at, stime)
            assert np.array_equal(st, rst)


def test_lense_blob_field_bound_holds_and_culling_is_invisible(oracle):
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_lense_field_bounds.restype = ctypes.c_longlong
    L.hostsim_check_lense_field_bounds.argtypes = [ctypes.c_longlong, ctypes.c_uint]
    assert L.hostsim_check_lense_field_bounds(3000000, 13) == 0
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    # between the fields (the bench camera), inside a field, skimming the blobs' tops, far above, and steep views
    views = [((3.4, 0.5, 6.1), (0.0, 0.0, 0.0)), ((1.0, -5.2, 1.0), (6.0, -4.0, 7.0)), ((0.0, 3.85, 2.0), (20.0, 3.9, 9.0)), ((2.0, 900.0, 3.0), (0.0, 0.0, 0.0)),
             ((0.5, 2.0, 0.5), (0.6, -30.0, 0.4)), ((0.2, -1.0, 8.0), (0.2, 40.0, 8.5)), ((5.0, 1500.0, 1.0), (5.0, 0.0, 1.2))]
    for eye, at in views:
        for stime, limits, svars in ((0.0, {}, None), (2.3, dict(max_cost_default=9, extension_lights=7), (1.5, -0.5, 9.0, 0.8))):
            f = oracle.default_frame("lense", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=stime)
            for k, v in limits.items():
                setattr(f, k, v)
            if svars:
                for i, x in enumerate(svars):
                    f.scene_var[i] = x
            ref, rst, _ = oracle.render("lense", f, stats=True)
            img, st = hostsim.render("lense", hostsim.frame_from_oracle(f))
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at, stime)
            assert np.array_equal(st, rst)


def test_terrain_octave_skip_is_invisible(oracle):
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_terrain_octave_skip.restype = ctypes.c_longlong
    L.hostsim_check_terrain_octave_skip.argtypes = [ctypes.c_longlong, ctypes.c_uint]
    assert L.hostsim_check_terrain_octave_skip(2000000, 17) == 0
    L.hostsim_check_terrain_height.restype = ctypes.c_longlong
    L.hostsim_check_terrain_height.argtypes = [ctypes.c_longlong, ctypes.c_uint, ctypes.POINTER(ctypes.c_double)]
    slack = ctypes.c_double(0)
    assert L.hostsim_check_terrain_height(3000000, 23, ctypes.byref(slack)) == 0  # fbm(p, p.y) >= p.y - 0.35: nothing above y = 0.35
    assert slack.value > 0.05
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    views = [((0.0, 2.0, -3.0), (0.0, 1.0, 0.0)), ((0.0, 9.0, -14.0), (0.0, 0.0, 0.0)), ((3.0, 0.7, 3.0), (-2.0, 0.4, -1.0)), ((0.5, 30.0, 0.5), (0.4, 0.0, 0.6)),
             ((12.0, 1.0, 0.0), (0.0, 0.5, 0.0)), ((2.0, 2000.0, 2.0), (0.0, 0.0, 0.0))]
    for eye, at in views:
        for levels in (2, 1, 5):
            f = oracle.default_frame("terrain", 80, 56, basis=oracle.camera_lookat(eye, at, fovy, np.float32(80.0 / 56.0)), stime=0.0)
            f.scene_var[0] = levels
            ref, rst, _ = oracle.render("terrain", f, stats=True)
            img, st = hostsim.render("terrain", hostsim.frame_from_oracle(f))
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at, levels)
            assert np.array_equal(st, rst)



def test_tree_border_skip_is_invisible(oracle):
    """the tree scene leaves the second half of its voronoi lattice out where a lower bound of the cell guard is not below the
    nearest object: the bound holds, and frames from inside, above and beside the forest keep their bits"""
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_tree_border_bound.restype = ctypes.c_longlong
    L.hostsim_check_tree_border_bound.argtypes = [ctypes.c_longlong, ctypes.c_uint, ctypes.POINTER(ctypes.c_double)]
    slack = ctypes.c_double(0)
    assert L.hostsim_check_tree_border_bound(3000000, 23, ctypes.byref(slack)) == 0
    assert slack.value >= 0.0
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    views = [((0.0, 2.0, -3.0), (0.0, 1.0, 0.0), 0.0), ((0.3, 1.2, 0.4), (2.0, 1.0, 3.0), 3.7), ((4.0, 0.4, 1.0), (0.0, 0.8, 0.0), 10.4),
             ((1.0, 9.0, 1.0), (1.2, 0.0, 1.1), 21.0), ((0.0, 1.9, 0.0), (10.0, 1.7, 4.0), 0.5), ((500.0, 1.5, -300.0), (510.0, 1.0, -290.0), 6.0)]
    for eye, at, stime in views:
        f = oracle.default_frame("tree", 72, 48, basis=oracle.camera_lookat(eye, at, fovy, np.float32(72.0 / 48.0)), stime=stime)
        ref, rst, _ = oracle.render("tree", f, stats=True)
        img, st = hostsim.render("tree", hostsim.frame_from_oracle(f))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at)
        assert np.array_equal(st, rst)


def test_cube_sea_escape_rule_changes_no_pixel(oracle):
    """FrameU::step_shortcuts with SceneCubeSea::ray_escapes (host build of the product's pipeline): same pixels, rays and
    hits as the oracle from every kind of view, fewer march evaluations"""
    import hostsim

    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    views = [((3.0, 4.5, 1.0), (6.0, 3.2, 4.0)), ((0.3, 0.4, 0.2), (2.0, 3.0, 1.5)), ((1.0, 9.0, 1.0), (1.2, 0.0, 1.1)),
             ((0.0, 2.0, 0.0), (8.0, 2.1, 3.0)), ((2.0, 3.7, 2.0), (9.0, 3.9, 2.5)), ((2.0, 3.655, 2.0), (9.0, 3.67, 2.5))]
    saved = 0
    for k, (eye, at) in enumerate(views):
        f = oracle.default_frame("cube_sea", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=0.4 * k)
        f.iter_count, f.max_cost_default = 128, 6
        ref, rst, _ = oracle.render("cube_sea", f, stats=True)
        hf = hostsim.frame_from_oracle(f)
        img, st = hostsim.render("cube_sea", hf)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(st, rst)
        hf.step_shortcuts = 1
        img, st = hostsim.render("cube_sea", hf)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at)
        assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]) and (st[..., 1] <= rst[..., 1]).all()
        saved += int(rst[..., 1].sum()) - int(st[..., 1].sum())
    assert saved > 0


def test_gems_bound_holds_and_culling_is_invisible(oracle):
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_gems_bound.restype = ctypes.c_longlong
    L.hostsim_check_gems_bound.argtypes = [ctypes.c_longlong, ctypes.c_uint, ctypes.POINTER(ctypes.c_double)]
    slack = ctypes.c_double(0)
    assert L.hostsim_check_gems_bound(6000000, 31, ctypes.byref(slack)) == 0
    assert slack.value > 0.005  # the 0.01 of slack is not eaten by rounding
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    views = [((2.5, 2.0, 0.3), (0.0, 1.0, 0.0)), ((0.2, 0.05, 0.1), (1.0, 1.0, 0.2)), ((1.05, 1.4, 0.0), (1.0, 0.0, 0.05)), ((0.0, 6.0, 0.01), (0.0, 1.0, 0.0)),
             ((1.4, 1.05, 0.2), (0.6, 1.0, -0.3))]
    for k, (eye, at) in enumerate(views):
        f = oracle.default_frame("gems", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=0.7 * k)
        f.max_cost_default, f.extension_lights = 9, 7
        ref, rst, _ = oracle.render("gems", f, stats=True)
        img, st = hostsim.render("gems", hostsim.frame_from_oracle(f))
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, at)
        assert np.array_equal(st, rst)


def test_fractal2_ball_and_gyroid_cull(oracle):
    """fractal2's boxes lie in the ball of radius 1.21 about the fold's centre (its escape rule and the cull of its fold), the shell
    scene's shells in the ball of radius 1.2161 about their cube's, and the gyroid's shape() is its cube's distance from 0.22 off the
    cube, bit for bit; frames from outside, inside and far away keep their bits"""
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_fractal2_gyroid_bounds.restype = ctypes.c_longlong
    L.hostsim_check_fractal2_gyroid_bounds.argtypes = [ctypes.c_longlong, ctypes.c_uint, ctypes.POINTER(ctypes.c_double)]
    slack = ctypes.c_double(0)
    assert L.hostsim_check_fractal2_gyroid_bounds(4000000, 11, ctypes.byref(slack)) == 0
    assert slack.value > 0.05  # the farthest box corner found stays that far inside the ball
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    for scene in ("fractal2", "gyroid", "shell", "sierpinski"):
        for eye, at in [((0.0, 2.0, -3.0), (0.0, 1.0, 0.0)), ((1.3, 1.9, 0.2), (0.0, 0.8, 0.0)), ((0.2, 0.9, 0.1), (2.0, 1.5, 1.0)), ((9.0, 7.0, -8.0), (0.0, 0.5, 0.0)),
                        ((0.0, 4.0, 0.01), (0.0, 0.0, 0.0))]:
            f = oracle.default_frame(scene, 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=0.9)
            ref, rst, _ = oracle.render(scene, f, stats=True)
            hf = hostsim.frame_from_oracle(f)
            img, st = hostsim.render(scene, hf)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (scene, eye)
            assert np.array_equal(st, rst), (scene, eye)
            hf.step_shortcuts = 1
            img, st = hostsim.render(scene, hf)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (scene, eye, "shortcuts")
            assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]) and (st[..., 1] <= rst[..., 1]).all()


def test_neon_rings_bound_and_escape_rule(oracle):
    """neon's rings lie on a sphere: their distance is never below | |p - c| - r1 | - r2 - 0.01 for any slider setting (the cull of
    their evaluation, the ball of the escape rule); frames with the sliders moved keep their bits with and without step shortcuts"""
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_neon_rings_bound.restype = ctypes.c_longlong
    L.hostsim_check_neon_rings_bound.argtypes = [ctypes.c_longlong, ctypes.c_uint, ctypes.POINTER(ctypes.c_double)]
    slack = ctypes.c_double(0)
    assert L.hostsim_check_neon_rings_bound(4000000, 13, ctypes.byref(slack)) == 0
    assert slack.value > 0.005  # the 0.01 of slack is not eaten by rounding
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    for k, (eye, at) in enumerate([((0.0, 2.0, -3.0), (0.0, 1.0, 0.0)), ((0.1, 2.05, 0.2), (0.0, 2.0, 2.75)), ((3.0, 0.5, 5.0), (0.0, 2.0, 1.0)), ((0.0, 9.0, 0.5), (0.0, 2.0, 1.0)),
                                   ((-2.0, 3.0, 6.0), (0.0, 2.0, 0.0))]):
        f = oracle.default_frame("neon", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=0.9)
        if k % 2:
            f.scene_var[0], f.scene_var[1], f.scene_var[2] = 1.9, 0.09, 0.03
        ref, rst, _ = oracle.render("neon", f, stats=True)
        hf = hostsim.frame_from_oracle(f)
        img, st = hostsim.render("neon", hf)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), eye
        assert np.array_equal(st, rst), eye
        hf.step_shortcuts = 1
        img, st = hostsim.render("neon", hf)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, "shortcuts")
        assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]) and (st[..., 1] <= rst[..., 1]).all()


def test_spiral_spring_bound_and_ball(oracle):
    """the hopping spring is never nearer than spring_lower_bound() says (the cull of its helix) and lies inside its bounding ball (the
    escape rule), at any time of the hop; frames keep their bits with and without step shortcuts"""
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_spiral_bounds.restype = ctypes.c_longlong
    L.hostsim_check_spiral_bounds.argtypes = [ctypes.c_longlong, ctypes.c_uint, ctypes.POINTER(ctypes.c_double)]
    slack = ctypes.c_double(0)
    assert L.hostsim_check_spiral_bounds(4000000, 19, ctypes.byref(slack)) == 0
    assert slack.value > 0.005
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    for k, (eye, at) in enumerate([((0.0, 2.0, -3.0), (0.0, 1.0, 0.0)), ((-2.93, 4.23, -4.22), (1.0, 2.5, 0.0)), ((0.3, 1.0, 0.2), (0.0, 5.0, 0.0)), ((6.0, 0.3, 6.0), (0.0, 2.0, 0.0)),
                                   ((0.0, 12.0, 0.1), (0.0, 0.0, 0.0))]):
        f = oracle.default_frame("spiral", 96, 64, basis=oracle.camera_lookat(eye, at, fovy, np.float32(1.5)), stime=0.55 * k + 0.1)
        ref, rst, _ = oracle.render("spiral", f, stats=True)
        hf = hostsim.frame_from_oracle(f)
        img, st = hostsim.render("spiral", hf)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), eye
        assert np.array_equal(st, rst), eye
        hf.step_shortcuts = 1
        img, st = hostsim.render("spiral", hf)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (eye, "shortcuts")
        assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]) and (st[..., 1] <= rst[..., 1]).all()


def test_lense_escape_distance_against_a_walk_along_the_ray(oracle):
    """SceneLense::escapes_from says from which distance on a ray meets nothing: a dense walk along 150 000 rays -- towards the sun
    (whose direction vector is shorter than 1 by up to dist_eps / |L|: the lever that broke a first version), towards lights
    between the fields, from inside and outside the slabs, the light ball anywhere -- finds the scene farther than 0.002 everywhere
    beyond that distance."""
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_lense_escape_rule.restype = ctypes.c_longlong
    L.hostsim_check_lense_escape_rule.argtypes = [ctypes.c_longlong, ctypes.c_uint, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_float)]
    fired = ctypes.c_longlong(0)
    witness = (ctypes.c_float * 9)()
    assert L.hostsim_check_lense_escape_rule(150000, 29, ctypes.byref(fired), witness) == 0, list(witness)
    assert fired.value > 30000


def test_every_escape_rule_against_a_walk_along_the_ray(oracle):
    """Every built-in scene's ray_escapes(): wherever it calls a ray gone -- from starts up to 45 away, along directions up to 6e-4
    shorter than unit vectors (a shadow ray towards a directional light), at random times and slider settings -- a dense walk along
    the rest of the ray finds the scene farther than 0.002, twice the largest dist_eps the library accepts."""
    import hostsim

    L = hostsim.lib()
    L.hostsim_check_escape_rule.restype = ctypes.c_longlong
    L.hostsim_check_escape_rule.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_uint, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_float)]
    L.hostsim_scene_count.restype = ctypes.c_int
    L.hostsim_scene_name.restype = ctypes.c_char_p
    L.hostsim_scene_name.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(20261107)
    with_rule = []
    for k in range(L.hostsim_scene_count()):
        scene = L.hostsim_scene_name(k).decode()
        table = oracle.var_table(scene)
        total = 0
        for rep in range(4):
            f = oracle.default_frame(scene, 64, 48, stime=float(rng.uniform(0, 40)))
            for name, mn, mx, start, _st, _v, slot in table:
                if slot >= 0 and rep:
                    f.scene_var[slot] = float(np.float32(rng.uniform(mn, mx)))
            hf = hostsim.frame_from_oracle(f)
            fired = ctypes.c_longlong(0)
            witness = (ctypes.c_float * 7)()
            bad = L.hostsim_check_escape_rule(scene.encode(), ctypes.byref(hf), 2000 if scene in ("tree", "terrain", "distortion") else 6000, 100 + rep, ctypes.byref(fired), witness)
            if bad == -1:
                break
            assert bad == 0, (scene, list(witness), [f.scene_var[i] for i in range(8)], f.stime)
            total += fired.value
        else:
            with_rule.append(scene)
            assert total > 300, (scene, total)
    assert len(with_rule) >= 20, with_rule
