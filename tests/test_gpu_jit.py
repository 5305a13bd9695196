"""GPU tier: scenes compiled at run time (sdfr_load_scene_source, hiprtc).

Parity anchor: the source text of a built-in scene, cut out of its header and compiled at run
time, must render the same bits as the oracle's restatement of that reference scene -- the
run-time path shares the pixel kernel body with the ahead-of-time path, so this pins it to the
same oracle."""
import numpy as np
import pytest

from jit_util import SCENES_DIR, aot_scene_source
from test_gpu_parity import H, W, _compare, _setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def renderer():
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    yield r
    r.close()


# run-time scenes get the plain IEEE sqrt / reciprocal / constant division unless their text opts into the
# exact fast forms (token SDFR_FAST_EXACT_MATH, sdfr_jit.cpp): on the verified domains -- where the built-in
# scenes stay -- both give the same bits, so both must match the oracle
@pytest.mark.parametrize("scene,struct_name,variables,fast_math", [
    ("fast_sphere", "SceneFastSphere", None, False),
    ("labyrinth", "SceneLabyrinth", None, False),
    ("labyrinth", "SceneLabyrinth", None, True),
    ("lense", "SceneLense", dict(mixing=0.8, zpos=9.0), True),
    ("light_shadows", "SceneLightShadows", None, False),
    ("tree", "SceneTree", None, True),        # WaveShare (per-wave LDS inside the scene text), waves_per_simd
    ("terrain", "SceneTerrain", None, False),
])
def test_built_in_scene_compiled_at_run_time_matches_oracle(renderer, oracle, scene, struct_name, variables, fast_math):
    f = _setup(renderer, oracle, scene, 0.75, variables=variables)
    table_aot = [(v.name, v.minval, v.maxval, v.start, v.step) for v in renderer.getVariableMap().values()]

    def scene_text(name):
        return ("// SDFR_FAST_EXACT_MATH: domains checked by the oracle census\n" if fast_math else "") + aot_scene_source(name)

    renderer.initShaderSource(scene + "_rt", scene_text(struct_name))
    assert renderer.currentScene() == scene + "_rt"
    # same variable table as the built-in scene (same tags in the text), then same values
    assert [(v.name, v.minval, v.maxval, v.start, v.step) for v in renderer.getVariableMap().values()] == table_aot
    for k, v in (variables or {}).items():
        assert renderer.setValue(k, v)
    assert _compare(renderer, oracle, scene, f, 1), "within tolerance but not bit-identical"
    # the debug-plane specialisation of the run-time kernel
    f2 = _setup(renderer, oracle, scene, 0.75, variables=dict(debug_ny=1.0, debug_y=0.5, **(variables or {})))
    renderer.initShaderSource(scene + "_rt", scene_text(struct_name))
    for k, v in dict(debug_ny=1.0, debug_y=0.5, **(variables or {})).items():
        assert renderer.setValue(k, v)
    assert _compare(renderer, oracle, scene, f2, 1)


def _pendulum_camera(sp):
    cam = sp.Camera()
    cam.SetEye((1.5, 2.5, -5.0))
    cam.SetLookat((0.0, 1.8, 0.0))
    cam.SetAspect(W / H)
    return cam


def test_example_scene_variables_and_reload(renderer):
    import sdf_playground_amd as sp

    renderer.initShaderSource("pendulum", SCENES_DIR + "/pendulum.scene.h")
    vm = renderer.getVariableMap()
    mine = [n for n in vm if not n.startswith("debug_") and n != "show_objects"]
    assert mine == ["radius", "rod", "swing"]                      # std::map order
    assert (vm["radius"].minval, vm["radius"].maxval, vm["radius"].start) == (np.float32(0.1), np.float32(0.8), np.float32(0.45))
    assert vm["rod"].step == np.float32(np.float32(2.5) - np.float32(0.5)) * np.float32(0.05)   # default step: 5 % of the range
    renderer.setParameters(0.4)
    cam = _pendulum_camera(sp)
    a = renderer.render(cam, W, H)
    b = renderer.render(cam, W, H)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.isfinite(a).all()
    assert a[..., :3].std() > 0.05                                  # a picture, not a constant
    renderer.setValue("radius", 0.8)
    c = renderer.render(cam, W, H)
    assert (c != a).any()
    # a scene that does not compile leaves the loaded one active (SceneManager.cpp:118-127)
    with pytest.raises(sp.SdfrError) as e:
        renderer.initShaderSource("broken", open(SCENES_DIR + "/pendulum.scene.h").read().replace("sd_sphere(", "sd_sfere("))
    assert "sd_sfere" in str(e.value)
    assert renderer.currentScene() == "pendulum"
    assert np.array_equal(renderer.render(cam, W, H).view(np.uint32), c.view(np.uint32))
    # edit and reload: a changed text gives a changed picture; variables restart from their defaults
    renderer.initShaderSource("pendulum", open(SCENES_DIR + "/pendulum.scene.h").read().replace("V3s(0.7f)", "V3s(0.1f)"))
    assert renderer.getVariableMap()["radius"].value == np.float32(0.45)
    d = renderer.render(cam, W, H)
    assert (d != a).any()
    # a built-in scene can be selected again afterwards
    renderer.initShader("fast_sphere")
    assert renderer.currentScene() == "fast_sphere"


def test_run_time_scene_renders_in_strips(renderer):
    import sdf_playground_amd as sp
    import torch

    renderer.initShaderSource("pendulum", SCENES_DIR + "/pendulum.scene.h")
    renderer.setParameters(1.1)
    cam = _pendulum_camera(sp)
    w, h = 200, 117
    full = renderer.render(cam, w, h)
    world = 3
    n = sp.strip_buffer_pixels(w, h, world)
    gathered = torch.empty((world, n, 4), dtype=torch.float32, device="cuda")
    for rank in range(world):
        renderer.renderStrips(w, h, rank, world, out=gathered[rank])
    out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    renderer.assembleStrips(w, h, world, gathered, out)
    renderer.sync()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), full.view(np.uint32))


def test_run_time_scene_in_the_persistent_launch(renderer):
    """a run-time scene is launched one wave per tile unless asked otherwise; asked, it runs the persistent launch of the
    built-in scenes (tile queue, waves that retire): same frame, same counters, at a size where waves take many tiles"""
    import sdf_playground_amd as sp
    import torch

    renderer.initShaderSource("pendulum", SCENES_DIR + "/pendulum.scene.h")
    renderer.setParameters(0.4)
    w, h = 2560, 1440
    cam = _pendulum_camera(sp)
    cam.SetAspect(w / h)
    try:
        want = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        renderer.render(cam, w, h, out=want)
        ref = renderer.getStats()
        renderer.setLaunchMode(sp.LAUNCH_PERSISTENT)
        got = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
        renderer.render(cam, w, h, out=got)
        s = renderer.getStats()
        assert (s.pixels, s.rays, s.march_evals, s.hits) == (ref.pixels, ref.rays, ref.march_evals, ref.hits) and s.pixels == w * h
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    finally:
        renderer.setLaunchMode(sp.LAUNCH_AUTO)
