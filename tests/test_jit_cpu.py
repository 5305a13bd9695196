"""CPU tier: scenes given as source text (the reference's edit-and-reload workflow,
SceneManager.cpp:102-133) compile for gfx950 without a device; errors come back as text."""
import pytest

import sdf_playground_amd as sp
from jit_util import SCENES_DIR, aot_scene_source


def test_example_scene_compiles_offline():
    ok, log = sp.check_scene_source(SCENES_DIR + "/pendulum.scene.h")
    assert ok, log


@pytest.mark.parametrize("struct_name", ["SceneFastSphere", "SceneLense", "SceneTree", "SceneTerrain"])  # the last two pass values round the wave through LDS
def test_built_in_scene_text_compiles_as_run_time_scene(struct_name):
    ok, log = sp.check_scene_source(aot_scene_source(struct_name))
    assert ok, log


def test_compile_error_is_reported_with_the_scene_line():
    src = open(SCENES_DIR + "/pendulum.scene.h").read().replace("sd_sphere(", "sd_sfere(")
    ok, log = sp.check_scene_source(src)
    assert not ok
    assert "sd_sfere" in log and "scene:" in log


def test_incomplete_scene_is_rejected():
    ok, log = sp.check_scene_source("struct Scene { static SDF_HD void prepare(FrameU &) {} };")
    assert not ok and "RayInv" in log


@pytest.mark.parametrize("bad", ["float f() { return VAR_x(min = a); }", "// VAR_ tags are explained in (the manual)", "VAR_a b(min=1)"])
def test_malformed_variable_tags_are_rejected_not_crashing(bad):
    ok, log = sp.check_scene_source("struct Scene {};\n" + bad)
    assert not ok and "VAR_" in log


def test_too_many_variables():
    src = "\n".join("float v%d(const FrameU &U) { return VAR_v%d(min = 0); }" % (i, i) for i in range(9))
    ok, log = sp.check_scene_source(src)
    assert not ok and "too many" in log
