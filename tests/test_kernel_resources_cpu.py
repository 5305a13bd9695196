"""Resource guard for the pixel kernels of the seven BASELINE configuration scenes (CPU tier: hipcc cross-compiles gfx950).

The headline kernel's speed hangs on things no functional test sees: how many vector registers the allocator took (waves
per SIMD), how much it spilled to scratch, whether a spill landed inside a march loop, and how many of a march loop's
three-source fma / fmac drew all their sources from one VGPR bank (half rate on gfx950, profiles/r03_bank_ubench.txt; the
allocator does not know, and an edit anywhere in the kernel re-rolls the draw: +-3 % on the headline).  Each
scene is compiled here with the options the build ships it with (buildlib.FLAGS + SCENE_FLAGS) and compared with the
committed table tests/golden/kernel_resources.json (tools/kernel_resources.py --write): an edit that makes any of these
worse fails here, and whoever makes it either finds another draw (buildlib.SCENE_FLAGS) or re-measures on the GPU and
records the new table on purpose."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def tables(tmp_path_factory):
    import kernel_resources as kr

    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here")
    with open(kr.BASELINE) as f:
        base = json.load(f)
    return kr.table(out_dir=str(tmp_path_factory.mktemp("isa"))), base


def test_every_configuration_scene_is_in_the_table(tables):
    import kernel_resources as kr

    now, base = tables
    assert sorted(now) == sorted(base) == sorted(kr.CONFIG_SCENES)


def test_registers_stay_within_the_scene_s_budget(tables):
    now, _base = tables
    for scene, r in now.items():
        # 512 vector registers per SIMD lane slot, shared by `waves_per_simd` waves, allocated in steps of 8
        assert r["vgprs"] <= r["vgpr_budget"], (scene, r["vgprs"], r["vgpr_budget"])
        assert r["occupancy"] >= r["waves_per_simd"], (scene, r["occupancy"])


def test_scratch_does_not_grow(tables):
    now, base = tables
    for scene, r in now.items():
        assert r["scratch_bytes"] <= base[scene]["scratch_bytes"], (scene, r["scratch_bytes"], base[scene]["scratch_bytes"])


def test_march_loops_are_free_of_scratch(tables):
    now, _base = tables
    for scene, r in now.items():
        assert r["march_loops"], scene
        assert r["march_loop_scratch"] == 0, (scene, r["march_loops"])


def test_one_bank_fma_draw_is_not_worse(tables):
    now, base = tables
    for scene, r in now.items():
        b = base[scene]
        assert len(r["march_loops"]) == len(b["march_loops"]), (scene, r["march_loops"], b["march_loops"])
        assert r["march_loop_one_bank_fma"] <= b["march_loop_one_bank_fma"], (scene, r["march_loops"], b["march_loops"])
        # the march loops did not grow either (instruction count is the other half of the issue-bound story)
        assert sum(m["valu"] for m in r["march_loops"]) <= sum(m["valu"] for m in b["march_loops"]) * 1.02 + 2, (scene, r["march_loops"], b["march_loops"])
