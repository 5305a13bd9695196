"""GPU tier: the C++ host mirror (include/sdfr.hpp) used from a plain g++ program linked
against libsdfr.so reproduces the golden fixtures."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror_reproduces_golden(tmp_path):
    import sdf_playground_amd as sp

    exe = str(tmp_path / "host_mirror")
    libdir = os.path.dirname(sp.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp"),
                    "-L" + libdir, "-lsdfr", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    for path in gu.golden_files():
        scene = gu.scene_of(path)
        if scene not in ("labyrinth", "lense", "cube_sea"):
            continue
        g = np.load(path)
        out = str(tmp_path / "img.raw")
        args = [exe, scene, repr(float(g["stime"])), str(int(g["width"])), out] + [repr(float(x)) for x in g["eye"]] + \
               [repr(float(x)) for x in g["target"]] + ["1" if bool(g["target_is_direction"]) else "0"]
        env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
        img = np.fromfile(out, np.float32).reshape(g["rgba"].shape)
        assert np.array_equal(img.view(np.uint32), g["rgba"].view(np.uint32)), scene
        if scene == "lense":
            assert "mixing=0.5" in r.stdout and "zpos=12.5" in r.stdout


def test_cpp_host_compiles_a_scene_file_at_run_time(tmp_path):
    """sdfr::SDFRenderer::initShaderSource from a plain g++ program: the text of the built-in
    fast_sphere scene, written to a file, must reproduce that scene's golden fixture."""
    import sdf_playground_amd as sp
    from jit_util import aot_scene_source

    exe = str(tmp_path / "host_reload")
    libdir = os.path.dirname(sp.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_reload.cpp"),
                    "-L" + libdir, "-lsdfr", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    src = str(tmp_path / "fast_sphere.scene.h")
    with open(src, "w") as fh:
        fh.write(aot_scene_source("SceneFastSphere"))
    path = [p for p in gu.golden_files() if gu.scene_of(p) == "fast_sphere"][0]
    g = np.load(path)
    assert not bool(g["target_is_direction"])
    out = str(tmp_path / "img.raw")
    args = [exe, src, repr(float(g["stime"])), str(int(g["width"])), out] + [repr(float(x)) for x in g["eye"]] + [repr(float(x)) for x in g["target"]]
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    img = np.fromfile(out, np.float32).reshape(g["rgba"].shape)
    assert np.array_equal(img.view(np.uint32), g["rgba"].view(np.uint32))


def test_cpp_host_shards_a_frame_over_its_gpus_without_pytorch(tmp_path):
    """tests/cpp/host_gather.cpp: one C++ process, one handle + one RCCL communicator per visible device
    (sdfr_comm_create_all), sdfr_render_gather_all; the assembled image equals a direct render (both formats,
    with and without private strips).  On a one-GPU box that is world 1; the same binary shards over N."""
    import sdf_playground_amd as sp

    exe = str(tmp_path / "host_gather")
    libdir = os.path.dirname(sp.LIB_PATH)
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_gather.cpp"),
                    "-L" + libdir, "-lsdfr", "-Wl,-rpath," + libdir, "-o", exe], check=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "200", "139"], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.count("identical") == 4 and "DIFFERENT" not in r.stdout


def test_cpp_host_loads_a_scene_in_the_reference_dialect(tmp_path):
    """sdfr::SDFRenderer::initShaderHlsl from a plain g++ program: scenes/pendulum.hlsl (map / map_light / map_background, the
    OBJECT and MATERIAL macros) renders the bits of its C++ twin scenes/pendulum.scene.h, with the VAR_ table the text declares."""
    import sdf_playground_amd as sp

    exe = str(tmp_path / "host_reload")
    libdir = os.path.dirname(sp.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_reload.cpp"),
                    "-L" + libdir, "-lsdfr", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    scenes = os.path.join(ROOT, "sdf_playground_amd", "scenes")
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    imgs = []
    for name in ("pendulum.hlsl", "pendulum.scene.h"):
        out = str(tmp_path / (name + ".raw"))
        r = subprocess.run([exe, os.path.join(scenes, name), "0.7", "160", out, "0.5", "2.0", "-6.0", "0.0", "1.5", "0.0"],
                           capture_output=True, text=True, env=env, timeout=240)
        assert r.returncode == 0, (name, r.returncode, r.stdout, r.stderr)
        assert "radius=0.45" in r.stdout and "rod=1.8" in r.stdout and "swing=0.7" in r.stdout
        imgs.append(np.fromfile(out, np.float32))
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    assert np.unique(imgs[0]).size > 1000
