"""GPU parity tests: the HIP kernels, called through the C ABI (libsdfr.so), against the CPU
oracle on identical camera / scene / variable inputs.

Bar (BASELINE.json north_star): <= 1e-4 per-channel L-infinity on the fp32 RGBA frame.
Because oracle and kernels share one arithmetic contract the frames are expected to be
bit-identical; the tests assert the stated tolerance and additionally require bit equality
of the per-pixel ray / march-evaluation / hit counters."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4
W, H = 240, 160
TH = 0.3


def _cameras(oracle):
    fovy = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)
    asp = np.float32(W) / np.float32(H)
    return {
        "fast_sphere": ("lookat", (0, 2, -3), (0, 1, 0)),
        "cube_sea": ("dir", (3 * math.cos(TH), 4.5, 3 * math.sin(TH)), (math.cos(TH + 0.6), -0.45, math.sin(TH + 0.6))),
        "labyrinth": ("dir", (1.5 * math.cos(TH), 5.0, 1.5 * math.sin(TH)), (math.cos(TH), -0.35, math.sin(TH))),
        "fractal": ("lookat", (2.2 * math.cos(TH), 1.6, 2.2 * math.sin(TH)), (0, 1, 0)),
        "lense": ("lookat", (7 * math.sin(-0.3), 0.5, 7 * math.cos(-0.3)), (0, 0, 0)),
        "gems": ("lookat", (2.5 * math.cos(TH), 2, 2.5 * math.sin(TH)), (0, 1, 0)),
        "light_shadows": ("lookat", (0, 5, -9), (0, 1, 0)),
        "cube": ("lookat", (2.5, 2.5, -3), (0, 1, 0)),
        "gyroid": ("lookat", (1.8, 1.6, -2.2), (0, 0, 0)),
        "basic_transparency": ("lookat", (2.0, 2.5, -4), (0, 2, 0)),
        "basic_clouds": ("lookat", (0, 2, -8), (0, 4, 0)),
        "coordinate_material": ("lookat", (3, 4, -5), (0, 2, 0)),
        "distortion": ("lookat", (0.8, 1.8, -2.5), (0, 1.5, 0)),
        "table": ("lookat", (2, 2, -3), (0, 1, 0)),
        "sierpinski": ("lookat", (1.2, 1.6, -2.2), (0, 1.2, 0)),
        "neon": ("lookat", (-2, 2.5, -3.5), (0, 2, 1)),
        "fractal2": ("lookat", (1.8, 1.8, -2.0), (0, 1, 0)),
        "shell": ("lookat", (-2.2, 1.8, -2.0), (0, 1, 0)),
        "spiral": ("lookat", (0, 4, -12), (0, 3, 0)),
        "terrain": ("lookat", (6, 5, -8), (0, 0, 0)),
        "tiling": ("lookat", (0, 4.5, -9), (0, 4, 0)),
        "tree": ("lookat", (0, 2.2, -4), (0, 1, 1)),
    }, fovy, asp


@pytest.fixture(scope="module")
def renderer():
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    yield r
    r.close()


def _setup(renderer, oracle, scene, stime, limits=None, variables=None):
    import sdf_playground_amd as sp

    cams, fovy, asp = _cameras(oracle)
    kind, eye, tgt = cams[scene]
    basis = (oracle.camera_lookat if kind == "lookat" else oracle.camera_direction)(eye, tgt, fovy, asp)
    f = oracle.default_frame(scene, W, H, basis=basis, stime=stime)
    renderer.initShader(scene)
    renderer.setParameters(stime)
    cam = sp.Camera()
    cam.SetEye(eye)
    (cam.SetLookat if kind == "lookat" else cam.SetDirection)(tgt)
    cam.SetFOVY(float(fovy))
    cam.SetAspect(float(asp))
    renderer.setCamera(cam)
    # the C++ host camera must reproduce the oracle's (and the reference's) basis bit for bit
    assert np.array_equal(renderer.getCameraBasis().view(np.uint32), basis.view(np.uint32))
    renderer.setLimits(iter_count=100, bounce_count=16, ray_count=8, light_count=8, range=100.0, max_cost_default=7, extension_lights=0,
                       extension_marble_reflection=0.0, dist_eps=0.0001, grad_eps=0.0001, reflect_eps=0.001, refract_eps=0.001, shadow_eps=0.0003)
    if limits:
        renderer.setLimits(**limits)
        for k, v in limits.items():
            setattr(f, k, v)
    if variables:
        table = {r[0]: r for r in oracle.var_table(scene)}
        for k, v in variables.items():
            assert renderer.setValue(k, v)
            slot = table[k][6]
            if slot >= 0:
                f.scene_var[slot] = v
            else:
                setattr(f, k, v)
    return f


def _compare(renderer, oracle, scene, f, schedule):
    renderer.setSchedule(schedule)
    img, st = renderer.render(None, W, H, pixel_stats=True)
    ref, rst, tot = oracle.render(scene, f, stats=True)
    assert not np.isnan(img).any()
    diff = np.abs(img.astype(np.float64) - ref.astype(np.float64)).max()
    assert diff <= TOL, "%s: L-inf %g" % (scene, diff)
    assert np.array_equal(st, rst), "%s: per-pixel ray/eval/hit counters differ" % scene
    s = renderer.getStats()
    assert (s.pixels, s.rays, s.march_evals, s.hits) == tuple(int(x) for x in tot)
    return np.array_equal(img.view(np.uint32), ref.view(np.uint32))


SCENES = ["fast_sphere", "cube_sea", "labyrinth", "fractal", "lense", "gems", "light_shadows", "cube", "gyroid", "basic_transparency",
          "basic_clouds", "coordinate_material", "distortion", "table", "sierpinski", "neon", "fractal2", "shell", "spiral", "terrain",
          "tiling", "tree"]


@pytest.mark.parametrize("schedule", [0, 1], ids=["wavefront", "pixel"])
@pytest.mark.parametrize("stime", [0.0, 1.25])
@pytest.mark.parametrize("scene", SCENES)
def test_scene_parity_reference_limits(renderer, oracle, scene, stime, schedule):
    f = _setup(renderer, oracle, scene, stime)
    assert _compare(renderer, oracle, scene, f, schedule), "within tolerance but not bit-identical"


@pytest.mark.parametrize("scene,limits", [
    ("fast_sphere", dict(iter_count=64, max_cost_default=2)),   # BASELINE config 1: no secondary rays
    ("cube_sea", dict(iter_count=128, max_cost_default=6)),     # config 2: exactly one reflection bounce
    ("labyrinth", dict(iter_count=256)),                         # config 3 (headline)
    ("fractal", dict(iter_count=512)),                           # config 4
    ("lense", dict(max_cost_default=9)),                         # config 5: recursion depth 4
    ("light_shadows", dict(ray_count=4, bounce_count=6)),       # queue overflow / bounce budget (Q4)
    ("gems", dict(light_count=0)),
    ("lense", dict(max_cost_default=9, extension_lights=7)),    # config 5 as worded: depth 4, 8 lights (queue overflow, Q4)
    ("gems", dict(max_cost_default=9, extension_lights=7)),
    ("light_shadows", dict(extension_lights=3)),                # extension slots overwrite the scene's own lights 1..3
    ("labyrinth", dict(iter_count=256, extension_marble_reflection=0.25)),  # config 3 as worded: reflective marble, 2 bounces
])
def test_scene_parity_extension_limits(renderer, oracle, scene, limits):
    f = _setup(renderer, oracle, scene, 0.5, limits=limits)
    for schedule in (0, 1):
        assert _compare(renderer, oracle, scene, f, schedule)


def test_variables_and_debug_views(renderer, oracle):
    # scene variables (lense) and the driver's debug plane / show_objects (pshader_sdf.hlsl:86-135)
    f = _setup(renderer, oracle, "lense", 0.25, variables=dict(xpos=1.5, ypos=-0.5, zpos=9.0, mixing=0.8))
    for schedule in (0, 1):
        assert _compare(renderer, oracle, "lense", f, schedule)
    f = _setup(renderer, oracle, "labyrinth", 0.25, variables=dict(debug_ny=1.0, debug_y=1.5, debug_scale=0.5))
    for schedule in (0, 1):
        assert _compare(renderer, oracle, "labyrinth", f, schedule)
    f = _setup(renderer, oracle, "fractal", 0.0, variables=dict(show_objects=0.0, debug_nx=0.3, debug_ny=1.0, debug_y=0.2))
    for schedule in (0, 1):
        assert _compare(renderer, oracle, "fractal", f, schedule)
    f = _setup(renderer, oracle, "coordinate_material", 0.0, variables=dict(spherical=1.0, thres=0.3, boxoffset=0.5))
    for schedule in (0, 1):
        assert _compare(renderer, oracle, "coordinate_material", f, schedule)
    f = _setup(renderer, oracle, "neon", 0.0, variables=dict(r1=1.4, spacing=0.15, red=2.0))
    for schedule in (0, 1):
        assert _compare(renderer, oracle, "neon", f, schedule)
    f = _setup(renderer, oracle, "tiling", 0.5, variables=dict(m1=0.7, m2=0.4, width=0.3, run_length=5.0, run_flip=3.0, flip_chance=0.8, truchet_width=0.15))
    for schedule in (0, 1):
        assert _compare(renderer, oracle, "tiling", f, schedule)
    f = _setup(renderer, oracle, "terrain", 0.0, variables=dict(levels=4.0))
    for schedule in (0, 1):
        assert _compare(renderer, oracle, "terrain", f, schedule)
    renderer.initShader("lense")  # reloading a scene resets its variables (Application.cpp:237)
    assert renderer.getVariableMap()["mixing"].value == 0.5


def test_fp16_target_and_device_output(renderer, oracle):
    import torch
    import sdf_playground_amd as sp

    f = _setup(renderer, oracle, "cube_sea", 0.0)
    ref, _, _ = oracle.render("cube_sea", f)
    out32 = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    out16 = torch.empty((H, W, 4), dtype=torch.float16, device="cuda")
    renderer.setStream(torch.cuda.current_stream().cuda_stream)
    renderer.render(None, W, H, out=out32)
    renderer.render(None, W, H, out=out16, fmt=sp.RGBA16F)
    renderer.sync()
    assert np.array_equal(out32.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # the reference's render target is R16G16B16A16_FLOAT (Postprocessing.cpp:23): fp32 rounded to half
    assert np.array_equal(out16.cpu().numpy().view(np.uint16), ref.astype(np.float16).view(np.uint16))
    renderer.setStream(0)


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (64, 8), (65, 9), (250, 131)])
def test_ragged_sizes(renderer, oracle, size):
    w, h = size
    renderer.initShader("fast_sphere")
    renderer.setParameters(0.0)
    import sdf_playground_amd as sp
    cam = sp.Camera()
    cam.SetAspect(float(np.float32(w) / np.float32(h)))
    renderer.setCamera(cam)
    renderer.setLimits(iter_count=100, bounce_count=16, ray_count=8, light_count=8, range=100.0, max_cost_default=7, extension_lights=0)
    f = oracle.default_frame("fast_sphere", w, h)
    ref, _, _ = oracle.render("fast_sphere", f)
    for schedule in (0, 1):
        renderer.setSchedule(schedule)
        img = renderer.render(None, w, h)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_strip_sharding_matches_full_frame(renderer, oracle, world):
    import torch
    import sdf_playground_amd as sp

    w, h = 200, 83  # ragged last strip
    f = _setup(renderer, oracle, "labyrinth", 0.75)
    cam = sp.Camera()
    renderer.setSchedule(0)
    full = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    renderer.render(None, w, h, out=full)
    n = sp.strip_buffer_pixels(w, h, world)
    assert n == sp.strip_buffer_pixels_host(w, h, world)
    gathered = torch.empty((world, n, 4), dtype=torch.float32, device="cuda")
    for rank in range(world):
        renderer.renderStrips(w, h, rank, world, gathered[rank])
    out = torch.empty_like(full)
    renderer.assembleStrips(w, h, world, gathered, out)
    renderer.sync()
    assert torch.equal(out, full)
    # the host statement of the layout (used by the CPU multi-rank tests) agrees
    assert np.array_equal(sp.assemble_strips_host(w, h, world, gathered.cpu().numpy()), full.cpu().numpy())
    # packed strips (13 bytes per pixel: rgb floats + the alpha flag as a byte) assemble to the same bits,
    # for both schedules and also for the fp16 strips
    nbytes = sp.strip_buffer_bytes(w, h, world, sp.STRIP_RGB32F_A8)
    assert nbytes == (13 * n + 3) // 4 * 4 and sp.strip_buffer_bytes(w, h, world, sp.RGBA32F) == 16 * n
    for schedule in (0, 1):
        renderer.setSchedule(schedule)
        packed = torch.full((world, nbytes), 0xAB, dtype=torch.uint8, device="cuda")
        for rank in range(world):
            renderer.renderStrips(w, h, rank, world, packed[rank], fmt=sp.STRIP_RGB32F_A8)
        out.zero_()
        renderer.assembleStrips(w, h, world, packed, out, fmt=sp.STRIP_RGB32F_A8)
        renderer.sync()
        assert torch.equal(out.view(torch.int32), full.view(torch.int32)), schedule
        # the host statement of the packed layout (CPU multi-rank test) agrees byte for byte
        for rank in range(world):
            assert np.array_equal(sp.pack_strip_host(gathered[rank].cpu().numpy()), packed[rank].cpu().numpy())
    half = torch.empty((world, n, 4), dtype=torch.float16, device="cuda")
    for rank in range(world):
        renderer.renderStrips(w, h, rank, world, half[rank], fmt=sp.RGBA16F)
    out16 = torch.empty((h, w, 4), dtype=torch.float16, device="cuda")
    renderer.assembleStrips(w, h, world, half, out16, fmt=sp.RGBA16F)
    renderer.sync()
    assert torch.equal(out16, full.to(torch.float16))


def test_error_behaviour(oracle):
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    with pytest.raises(sp.SdfrError) as e:
        r.render(None, 16, 16)  # no scene: SDFRenderer::render returns false (SDFRenderer.cpp:70-73)
    assert e.value.code == -4
    with pytest.raises(sp.SdfrError) as e:
        r.initShader("no_such_scene")
    assert e.value.code == -2
    r.initShader("lense")
    assert r.setValue("does_not_exist", 1.0) is False  # ignored (ShaderUtil.cpp:234-240)
    with pytest.raises(sp.SdfrError):
        r.setLimits(ray_count=9)
    names = list(r.getVariableMap().keys())
    assert names == [row[0] for row in oracle.var_table("lense")]
    for row in oracle.var_table("lense"):
        v = r.getVariableMap()[row[0]]
        assert (v.minval, v.maxval, v.start, v.step, v.value) == tuple(np.float32(x) for x in row[1:6])
    r.close()


def test_kernels_reproduce_golden_fixtures(renderer):
    """Committed fixtures (tests/golden, tools/make_golden.py): no oracle in the loop."""
    import golden_util as gu
    import sdf_playground_amd as sp

    for path in gu.golden_files():
        g = np.load(path)
        scene = gu.scene_of(path)
        renderer.initShader(scene)
        renderer.setParameters(float(g["stime"]))
        renderer.setLimits(iter_count=int(g["iter_count"]), bounce_count=int(g["bounce_count"]), ray_count=int(g["ray_count"]),
                           light_count=int(g["light_count"]), range=float(g["range"]), max_cost_default=int(g["max_cost_default"]))
        b = g["basis"]
        renderer.setCameraBasis(b[0], b[1], b[2], b[3])
        w, h = int(g["width"]), int(g["height"])
        for schedule in (0, 1):
            renderer.setSchedule(schedule)
            img, st = renderer.render(None, w, h, pixel_stats=True)
            assert np.abs(img.astype(np.float64) - g["rgba"]).max() <= TOL
            assert np.array_equal(img.view(np.uint32), g["rgba"].view(np.uint32)), path
            assert np.array_equal(st, g["stats"].astype(np.uint32)), path


def test_variable_tables_of_all_scenes(oracle):
    """Names, order (std::map) and defaults of every scene's VAR_ table: C++ host parser of the
    product vs the oracle's restatement (which the CPU tier checks against the reference's
    scene files)."""
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    assert sp.scene_names() == oracle.scene_names()
    for scene in sp.scene_names():
        r.initShader(scene)
        got = [(v.name, v.minval, v.maxval, v.start, v.step, v.value) for v in r.getVariableMap().values()]
        want = [(row[0],) + tuple(float(np.float32(x)) for x in row[1:6]) for row in oracle.var_table(scene)]
        assert got == want, scene
    r.close()


def _frames256():
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames256.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("case", _frames256()["frames"], ids=lambda c: "%s-t%g%s" % (c["scene"], c["stime"], "-cfg1" if c["limits"] else ""))
def test_kernels_reproduce_256x256_digests(renderer, case):
    """The committed 256x256 digests (tools/make_golden256.py), no oracle involved: start-up camera
    of the reference, both schedules."""
    import hashlib

    import sdf_playground_amd as sp

    renderer.initShader(case["scene"])
    renderer.setParameters(case["stime"])
    # every limit, also the extensions and the epsilons: the handle is shared by the module's tests, and whatever ran before this one
    # (which depends on the -k selection) must not leak into the digests
    renderer.setLimits(iter_count=100, bounce_count=16, ray_count=8, light_count=8, range=100.0, max_cost_default=7, extension_lights=0,
                       extension_marble_reflection=0.0, dist_eps=0.0001, grad_eps=0.0001, reflect_eps=0.001, refract_eps=0.001, shadow_eps=0.0003)
    renderer.resetVariables()
    renderer.setLimits(**case["limits"])
    cam = sp.Camera()          # Application.cpp:214-224
    cam.SetAspect(1.0)
    for schedule in (sp.SCHEDULE_PIXEL, sp.SCHEDULE_WAVEFRONT):
        renderer.setSchedule(schedule)
        img, st = renderer.render(cam, 256, 256, pixel_stats=True)
        assert hashlib.sha256(img.tobytes()).hexdigest() == case["rgba_sha256"], (case["scene"], schedule)
        assert hashlib.sha256(st.tobytes()).hexdigest() == case["stats_sha256"]
        s = renderer.getStats()
        assert [s.pixels, s.rays, s.march_evals, s.hits] == case["totals"]


@pytest.mark.parametrize("world,split", [(2, (5, 16)), (4, (1, 4)), (8, (3, 16)), (1, (3, 4)), (3, (15, 16))])
def test_private_strips_plus_shared_strips_make_the_full_frame(renderer, oracle, world, split):
    """Unequal shares: the root renders the private strips straight into the image, all ranks share
    the rest (sdfr_set_strip_split); emulated on one GPU, both schedules, packed transport."""
    import torch
    import sdf_playground_amd as sp

    w, h = 200, 139
    _setup(renderer, oracle, "labyrinth", 0.75)
    renderer.setStripSplit(0, 1)
    renderer.setSchedule(1)
    full = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    renderer.render(None, w, h, out=full)
    try:
        for schedule in (1, 0):
            renderer.setSchedule(schedule)
            renderer.setStripSplit(*split)
            n = sp.strip_buffer_pixels(w, h, world, split)
            assert n == sp.strip_buffer_pixels_host(w, h, world, split)
            nbytes = sp.strip_buffer_bytes(w, h, world, sp.STRIP_RGB32F_A8, split)
            packed = torch.full((world, nbytes), 0xCD, dtype=torch.uint8, device="cuda")
            for rank in range(world):
                renderer.renderStrips(w, h, rank, world, packed[rank], fmt=sp.STRIP_RGB32F_A8)
            out = torch.full((h, w, 4), -7.0, dtype=torch.float32, device="cuda")
            renderer.renderPrivateStrips(w, h, out)
            renderer.sync()
            # exactly the private rows are written by the private render
            priv = sp.private_rows_host(h, split)
            mask = torch.zeros(h, dtype=torch.bool, device="cuda")
            mask[priv] = True
            assert torch.equal(out[mask].view(torch.int32), full[mask].view(torch.int32))
            assert bool((out[~mask] == -7.0).all())
            renderer.assembleStrips(w, h, world, packed, out, fmt=sp.STRIP_RGB32F_A8)
            renderer.sync()
            assert torch.equal(out.view(torch.int32), full.view(torch.int32)), (schedule, world, split)
            # the host statement of the layout agrees
            unpacked = np.stack([sp.unpack_strip_host(packed[r].cpu().numpy(), n) for r in range(world)])
            img = sp.assemble_strips_host(w, h, world, unpacked, split, image=np.where(mask.cpu().numpy()[:, None, None], full.cpu().numpy(), 0).astype(np.float32))
            assert np.array_equal(img, full.cpu().numpy())
    finally:
        renderer.setStripSplit(0, 1)
        renderer.setSchedule(1)


@pytest.mark.parametrize("scene", ["labyrinth", "fast_sphere", "light_shadows", "lense"])
def test_launch_modes_give_the_same_frame(renderer, oracle, scene):
    """one wave per tile and the persistent launch (resident waves pulling tiles from the cursors, guided
    claims) against the oracle, on a ragged frame large enough that waves take many tiles, twice in a row
    (the cursors must come back to zero), pixels and counters"""
    import torch
    import sdf_playground_amd as sp

    f = _setup(renderer, oracle, scene, 0.75)
    try:
        assert _compare(renderer, oracle, scene, f, 1)  # AUTO
        w, h = 1003, 517
        want = None
        for mode in (sp.LAUNCH_PER_TILE, sp.LAUNCH_PERSISTENT, sp.LAUNCH_AUTO):
            renderer.setLaunchMode(mode)
            assert _compare(renderer, oracle, scene, f, 1), mode
            for rep in range(2):
                img = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
                pst = torch.empty((h, w, 3), dtype=torch.int32, device="cuda")
                renderer.render(None, w, h, out=img, pixel_stats=pst)
                s = renderer.getStats()
                got = (img.cpu().numpy().view(np.uint32), pst.cpu().numpy(), (s.pixels, s.rays, s.march_evals, s.hits))
                assert got[2][0] == w * h and int(got[1][..., 0].sum()) == got[2][1] and int(got[1][..., 1].sum()) == got[2][2]
                if want is None:
                    want = got
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2], (mode, rep)
    finally:
        renderer.setLaunchMode(sp.LAUNCH_AUTO)


@pytest.mark.parametrize("scene", ["fast_sphere", "labyrinth"])
def test_persistent_launch_at_full_size_loses_no_tile(renderer, oracle, scene):
    """3840 x 2160: 129 600 tiles for ~7 000 resident waves.  With cheap tiles (fast_sphere) a wave claims up to eight per
    atomic, and a wave retires after a number of tiles to make room for a younger one -- never while it still holds
    claimed tiles (a wave that did left holes in the frame).  Same bits as one wave per tile, every pixel counted, twice."""
    import torch
    import sdf_playground_amd as sp

    _setup(renderer, oracle, scene, 0.75)
    w, h = 3840, 2160
    cam = sp.Camera()
    cam.SetEye((1.2, 5.0, 0.4))
    cam.SetDirection((0.9, -0.35, 0.3))
    cam.SetAspect(w / h)
    renderer.setCamera(cam)
    try:
        renderer.setLaunchMode(sp.LAUNCH_PER_TILE)
        want = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
        renderer.render(None, w, h, out=want)
        ref = renderer.getStats()
        assert ref.pixels == w * h
        renderer.setLaunchMode(sp.LAUNCH_PERSISTENT)
        for rep in range(2):
            img = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
            renderer.render(None, w, h, out=img)
            s = renderer.getStats()
            assert (s.pixels, s.rays, s.march_evals, s.hits) == (ref.pixels, ref.rays, ref.march_evals, ref.hits), rep
            assert torch.equal(img.view(torch.int32), want.view(torch.int32)), rep
    finally:
        renderer.setLaunchMode(sp.LAUNCH_AUTO)


def test_frame_sequences_keep_their_bits(renderer, oracle):
    """a persistent launch learns the order of its tile rows from the frame before (row feedback): a moving camera, a change
    of size in between (the learned order is for another number of rows: ignored), a change of scene -- every frame equals the
    same frame rendered one wave per tile, pixels and counters"""
    import torch
    import bench
    import sdf_playground_amd as sp

    try:
        for scene, config in (("labyrinth", "3"), ("cube_sea", "2")):
            _setup(renderer, oracle, scene, 0.75)
            renderer.setLimits(**bench.CONFIGS[config]["limits"])
            for k in range(6):
                w, h = ((1920, 1080), (1920, 1080), (648, 360), (1920, 1080), (1000, 562), (1920, 1080))[k]
                cam, stime = bench.make_camera(k, w, h, config)
                renderer.setParameters(stime)
                got = {}
                for mode in (sp.LAUNCH_PERSISTENT, sp.LAUNCH_PER_TILE):
                    renderer.setLaunchMode(mode)
                    img = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
                    renderer.render(cam, w, h, out=img)
                    s = renderer.getStats()
                    got[mode] = (img, (s.pixels, s.rays, s.march_evals, s.hits))
                assert got[sp.LAUNCH_PERSISTENT][1] == got[sp.LAUNCH_PER_TILE][1] and got[sp.LAUNCH_PER_TILE][1][0] == w * h, (scene, k)
                assert torch.equal(got[sp.LAUNCH_PERSISTENT][0].view(torch.int32), got[sp.LAUNCH_PER_TILE][0].view(torch.int32)), (scene, k)
    finally:
        renderer.setLaunchMode(sp.LAUNCH_AUTO)


def test_registered_host_target(renderer, oracle):
    """sdfr_register_host_target: a page-locked host image is filled with the same bits, frame after frame; after
    unregistering, ordinary host destinations still work"""
    f = _setup(renderer, oracle, "labyrinth", 0.75)
    ref, _, _ = oracle.render("labyrinth", f)
    host = np.full((H, W, 4), 7.0, np.float32)
    renderer.registerHostTarget(host)
    try:
        for _ in range(3):
            host[...] = 7.0
            renderer.render(None, W, H, out=host)
            assert np.array_equal(host.view(np.uint32), ref.view(np.uint32))
    finally:
        renderer.registerHostTarget(None)
    assert np.array_equal(renderer.render(None, W, H).view(np.uint32), ref.view(np.uint32))


def test_two_frames_in_flight_inside_one_handle(renderer, oracle):
    """sdfr_set_frames_in_flight(2): render() alternates between two internal streams and workspaces.  A sequence of frames
    (moving camera, a change of size, a change of scene, the same buffer twice in a row) gives, frame for frame, the pixels
    and counters of the same sequence rendered one frame at a time."""
    import torch
    import bench
    import sdf_playground_amd as sp

    def sequence(pipelined):
        renderer.setFramesInFlight(2 if pipelined else 1)
        frames = []
        try:
            for scene, config in (("labyrinth", "3"), ("cube_sea", "2"), ("gems", "5g")):
                _setup(renderer, oracle, scene, 0.75)
                renderer.setLimits(**bench.CONFIGS[config]["limits"])
                bufs = {}
                for k in range(7):
                    w, h = ((960, 540), (960, 540), (648, 360), (960, 540), (960, 540), (500, 281), (960, 540))[k]
                    cam, stime = bench.make_camera(k, w, h, config)
                    renderer.setParameters(stime)
                    # two images per size, taken in turn -- except frames 3 and 4, which go into the SAME image (the second one
                    # has to wait for the first: sdfr.h)
                    pair = bufs.setdefault((w, h), [torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda") for _ in range(2)])
                    img = pair[0] if k in (3, 4) else pair[k & 1]
                    renderer.render(cam, w, h, out=img)
                    s = renderer.getStats()  # waits for this frame only
                    frames.append((scene, k, img.clone(), (s.pixels, s.rays, s.march_evals, s.hits)))
            renderer.sync()
        finally:
            renderer.setFramesInFlight(1)
        return frames

    one = sequence(False)
    two = sequence(True)
    assert len(one) == len(two) == 21
    for a, b in zip(one, two):
        assert a[3] == b[3] and a[3][0] > 0, (a[0], a[1], a[3], b[3])
        assert torch.equal(a[2].view(torch.int32), b[2].view(torch.int32)), (a[0], a[1])
    # without a wait in between, frames in flight still land in their own images
    renderer.setFramesInFlight(2)
    try:
        _setup(renderer, oracle, "labyrinth", 0.75)
        renderer.setLimits(**bench.CONFIGS["3"]["limits"])
        imgs = [torch.full((540, 960, 4), -1.0, dtype=torch.float32, device="cuda") for _ in range(4)]
        for k in range(4):
            cam, stime = bench.make_camera(k, 960, 540, "3")
            renderer.setParameters(stime)
            renderer.render(cam, 960, 540, out=imgs[k])
        side = torch.cuda.Stream()
        renderer.waitFrame(side.cuda_stream)  # the side stream waits on the device for frame 3
        with torch.cuda.stream(side):
            copy3 = imgs[3].clone()
        renderer.sync()
        side.synchronize()
        for k in range(4):
            ref = [f for f in one if f[0] == "labyrinth" and f[1] == k][0] if k not in (2,) else None
            if ref is not None and ref[2].shape == imgs[k].shape:
                assert torch.equal(ref[2].view(torch.int32), imgs[k].view(torch.int32)), k
        assert torch.equal(copy3.view(torch.int32), imgs[3].view(torch.int32))
        # the same image twice in a row, nothing waited for in between: the second frame is what stays
        same = torch.full((540, 960, 4), -1.0, dtype=torch.float32, device="cuda")
        for k in (3, 4):
            cam, stime = bench.make_camera(k, 960, 540, "3")
            renderer.setParameters(stime)
            renderer.render(cam, 960, 540, out=same)
        renderer.sync()
        ref4 = [f for f in one if f[0] == "labyrinth" and f[1] == 4][0]
        assert torch.equal(ref4[2].view(torch.int32), same.view(torch.int32))
    finally:
        renderer.setFramesInFlight(1)
