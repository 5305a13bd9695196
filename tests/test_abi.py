"""CPU tier: the C-ABI library builds for gfx950, loads, exports every symbol that
include/sdfr.h declares, and refuses to work without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import sdf_playground_amd as sp

    sp.build()
    return sp.load_library()


def test_header_symbols_are_exported(lib):
    import sdf_playground_amd as sp

    header = open(os.path.join(ROOT, "include", "sdfr.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|void|int64_t|const char \*)\s*(sdfr_[a-z_0-9]+)\(", header, re.M)))
    assert declared == sorted(sp.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_scene_list_matches_reference_stems(lib):
    import sdf_playground_amd as sp

    assert sp.scene_names() == ["fast_sphere", "cube_sea", "labyrinth", "fractal", "lense", "gems", "light_shadows", "cube", "gyroid",
                                "basic_transparency", "basic_clouds", "coordinate_material", "distortion", "table", "sierpinski", "neon",
                                "fractal2", "shell", "spiral", "terrain", "tiling", "tree"]
    # every name is a stem of the reference's scenes directory (SceneManager.cpp:142-158)
    ref = "/root/reference/Engine/shader/scenes"
    if os.path.isdir(ref):
        stems = {f[len("sdf_scene_"):-5] for f in os.listdir(ref) if f.endswith(".hlsl")}
        assert set(sp.scene_names()) == stems  # all 22 scenes of the reference


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    import sdf_playground_amd as sp

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sp.SdfrError) as e:
        sp.SDFRenderer(0)
    assert e.value.code in (-6, -5)


def test_strip_layout_host_statement():
    import numpy as np
    import sdf_playground_amd as sp

    w, h, world = 5, 29, 3
    n = sp.strip_buffer_pixels_host(w, h, world)
    assert n == 2 * 8 * w  # 4 strips -> 2 per rank
    img = np.arange(h * w * 2, dtype=np.float32).reshape(h, w, 2)
    gathered = np.zeros((world, n // w, w, 2), np.float32)
    for rank in range(world):
        for k, row in enumerate(sp.strip_rows_of_rank(h, rank, world)):
            strip = row // 8
            gathered[rank, (strip // world) * 8 + row % 8] = img[row]
    assert np.array_equal(sp.assemble_strips_host(w, h, world, gathered.reshape(world, n, 2)), img)


def test_no_cpp_exception_crosses_the_c_boundary(lib):
    """every C entry point with a body runs inside guarded() (csrc/sdfr_handle.h): an exception thrown inside the library -- the
    self-test throws a std::runtime_error, a std::bad_alloc and a non-std object on purpose -- comes back as SDFR_ERR_INTERNAL;
    a ctypes (or C) caller would otherwise die of std::terminate.  Needs no device."""
    for what in (0, 1, 2):
        assert lib.sdfr_selftest_exception(None, what) == -9
    assert lib.sdfr_selftest_exception(None, 3) == 0
