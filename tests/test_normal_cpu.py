"""CPU tier: the scene ABI's map_normal / NormalOutput (sdf_structs.hlsl:39-52, pshader_sdf.hlsl:318-330 and :520) and the
driver's five epsilons (pshader_sdf.hlsl:31-35) as run-time limits.

No scene of the reference fills map_normal in, so a test scene does ("normal_test": oracle/test_scenes.h; the library's
diagnostic scene of that name, csrc/sdfr_scene_debug.h): the oracle against what the reference's comments promise
("larger than usual values lead to rounded corners", "uses the user generated normal instead of computing it"), and the
host build of the product's pipeline stages against the oracle, bit for bit."""
import numpy as np
import pytest

W, H = 120, 80
FOVY = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
CAMS = [((0.2, 2.2, -4.6), (0.0, 0.5, -0.2)), ((3.2, 1.1, -2.4), (1.2, 0.5, -0.3)), ((-3.4, 1.5, -1.6), (-1.2, 0.6, 0.1)), ((0.8, 5.0, -0.9), (0.6, 0.0, -0.5))]
EPS_DEFAULT = dict(dist_eps=0.0001, grad_eps=0.0001, reflect_eps=0.001, refract_eps=0.001, shadow_eps=0.0003)
EPS_OTHER = dict(dist_eps=0.0005, grad_eps=0.002, reflect_eps=0.004, refract_eps=0.0025, shadow_eps=0.0011)


def _frame(oracle, scene, cam, stime=0.4, w=W, h=H, **kw):
    f = oracle.default_frame(scene, w, h, basis=oracle.camera_lookat(cam[0], cam[1], FOVY, np.float32(w) / np.float32(h)), stime=stime)
    slots = {r[0]: r[6] for r in oracle.var_table(scene)}
    for k, v in kw.items():
        if k in slots and slots[k] >= 0:
            f.scene_var[slots[k]] = v
        else:
            setattr(f, k, v)
    return f


def test_default_frame_carries_the_reference_epsilons(oracle):
    f = oracle.default_frame("normal_test", 8, 8)
    for k, v in EPS_DEFAULT.items():
        assert getattr(f, k) == np.float32(v), k
    assert [r[0] for r in oracle.var_table("normal_test") if r[6] >= 0] == ["analytic", "round"]  # std::map order
    assert "normal_test" not in oracle.scene_names()


def test_wider_sample_spacing_rounds_the_corners(oracle):
    """the drum wears MATERIAL_NORMAL2 (colour = |normal|): on its flat lid the sampled normal is (0, 1, 0) whatever the
    spacing; within `round` of the rim the forward differences straddle the edge and the normal turns over gradually"""
    cam = ((2.6, 2.6, -2.2), (1.5, 0.45, -0.3))  # looking down on the drum
    blended = []
    for rnd in (0.0001, 0.01, 0.04):
        img, st, _ = oracle.render("normal_test", _frame(oracle, "normal_test", cam, round=rnd), stats=True)
        drum = (img[..., 3] == 0.0) & (st[..., 2] >= 1)  # NORMAL2 switches tone mapping off; nothing else in this scene does
        assert drum.sum() > 400
        n = img[drum][:, :3]
        lid = (n[:, 1] == 1.0) & (n[:, 0] == 0.0) & (n[:, 2] == 0.0)
        assert lid.sum() > 150  # exact on the flat part
        blended.append(int(((n[:, 1] > 0.15) & (n[:, 1] < 0.99) & (np.maximum(n[:, 0], n[:, 2]) > 0.15)).sum()))
    assert blended[0] == 0 and blended[1] >= 5 and blended[2] > 3 * blended[1], blended


def test_the_scene_s_own_normal_replaces_the_sampled_one(oracle):
    """analytic = 1: map_normal hands the ball's normal over (use_normal), the three forward-difference evaluations are not
    made; the picture moves only on the ball and what the ball sheds light or shadow on, and only a little (the sampled
    normal of a sphere is accurate to ~1e-3), while the march counters do not move at all"""
    cam = CAMS[2]
    a, sa, _ = oracle.render("normal_test", _frame(oracle, "normal_test", cam, analytic=1.0), stats=True)
    b, sb, _ = oracle.render("normal_test", _frame(oracle, "normal_test", cam, analytic=0.0), stats=True)
    assert np.array_equal(sa[..., 0], sb[..., 0]) or (sa[..., 0] != sb[..., 0]).mean() < 0.01  # a grazing shadow ray may flip
    changed = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
    assert 300 < changed.sum() < 0.5 * W * H
    assert np.abs(a - b)[changed].max() < 0.05
    # a ball pixel by construction: centre (-1.6, 0.7, 0.2) projected -- the changed set covers it
    ys, xs = np.nonzero(changed)
    assert xs.min() < W * 0.6 and ys.max() > H * 0.3


@pytest.mark.parametrize("cam", range(len(CAMS)))
@pytest.mark.parametrize("extra", [{}, dict(round=0.04), dict(analytic=0.0, round=0.0001), dict(max_cost_default=9, extension_lights=7),
                                   dict(EPS_OTHER), dict(debug_ny=1.0, debug_y=0.45)])
def test_normal_test_host_build_vs_oracle(oracle, cam, extra):
    import hostsim

    f = _frame(oracle, "normal_test", CAMS[cam], **extra)
    ref, rst, _ = oracle.render("normal_test", f, stats=True)
    for shortcuts in (0, 1):
        hf = hostsim.frame_from_oracle(f)
        hf.step_shortcuts = shortcuts
        img, st = hostsim.render("normal_test", hf)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (cam, extra, shortcuts)
        assert np.array_equal(st, rst)


def test_epsilons_reach_every_use(oracle):
    """each epsilon moves the picture of a scene that exercises it (so none of the five is a dead parameter), in the oracle"""
    cam, sea = ((1.5, 3.0, -4.0), (0.0, 1.0, 0.0)), ((3.0, 4.5, 1.0), (6.0, 3.2, 4.0))
    cases = [("dist_eps", "fast_sphere", 0.0009, cam), ("grad_eps", "labyrinth", 0.01, cam), ("reflect_eps", "cube_sea", 0.05, sea),
             ("refract_eps", "gems", 0.05, cam), ("shadow_eps", "light_shadows", 0.02, cam)]
    for name, scene, value, view in cases:
        a, _, _ = oracle.render(scene, _frame(oracle, scene, view, w=72, h=48))
        b, _, _ = oracle.render(scene, _frame(oracle, scene, view, w=72, h=48, **{name: value}))
        assert (a.view(np.uint32) != b.view(np.uint32)).any(), name


def test_every_scene_at_other_epsilons_host_build_vs_oracle(oracle):
    """all 22 scenes (+ the two diagnostic ones) with all five epsilons off their defaults: the product's stages read the
    run-time values wherever the oracle's literal restatement reads the reference's constants"""
    import hostsim

    rng = np.random.default_rng(77)
    for scene in oracle.scene_names() + ["debug_materials", "normal_test"]:
        heavy = scene in ("tree", "terrain", "distortion", "tiling")
        w, h = (40, 28) if heavy else (64, 44)
        eye = (float(rng.uniform(-5, 5)), float(rng.uniform(0.4, 5)), float(rng.uniform(-6, -2)))
        f = _frame(oracle, scene, (eye, (0.0, 1.0, 0.0)), stime=0.8, w=w, h=h, **EPS_OTHER)
        ref, rst, _ = oracle.render(scene, f, stats=True)
        for shortcuts in (0, 1):
            hf = hostsim.frame_from_oracle(f)
            hf.step_shortcuts = shortcuts
            img, st = hostsim.render(scene, hf)
            same = np.array_equal(img.view(np.uint32), ref.view(np.uint32)) or np.array_equal(img, ref, equal_nan=True)
            assert same, (scene, shortcuts, int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()))
            assert np.array_equal(st[..., 0], rst[..., 0]) and np.array_equal(st[..., 2], rst[..., 2]), (scene, shortcuts)
            assert np.array_equal(st[..., 1], rst[..., 1]) if not shortcuts else (st[..., 1] <= rst[..., 1]).all()
