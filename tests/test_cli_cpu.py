"""CPU tier: the VAR_ tag reader of the command-line front end agrees with the oracle's
restatement of ShaderVariableManager::parseFile, on synthetic text and -- when the reference
checkout is present -- on every scene file of the reference (read as text: metadata only)."""
import glob
import os

import numpy as np


def test_var_tag_reader_matches_parser_restatement(oracle):
    from sdf_playground_amd.cli import parse_var_tags

    text = "float a = VAR_foo(); VAR_bar(min = -4, max = 4, step = 0.1) VAR_q(min=1,max=3,start=1.5) VAR_t(min = 0, max = 10, steps = 1) VAR_x(min=0,max=1) VAR_x(min=0,max=4)"
    mine = parse_var_tags(text)
    ref = oracle.parse_vars(text)
    assert list(mine) == list(ref)
    for k in ref:
        assert tuple(np.float32(v) for v in mine[k]) == tuple(np.float32(v) for v in ref[k][:4]), k


def test_scene_files_of_the_reference_declare_what_the_library_serves(oracle):
    from sdf_playground_amd.cli import parse_var_tags

    scenes_dir = "/root/reference/Engine/shader/scenes"
    if not os.path.isdir(scenes_dir):
        import pytest
        pytest.skip("reference checkout not present")
    driver = parse_var_tags(open("/root/reference/Engine/shader/pshader_sdf.hlsl").read())
    assert sorted(driver) == ["debug_nx", "debug_ny", "debug_nz", "debug_scale", "debug_x", "debug_y", "debug_z", "show_objects"]
    served = set(oracle.scene_names())
    for path in sorted(glob.glob(os.path.join(scenes_dir, "*.hlsl"))):
        stem = os.path.basename(path)[len("sdf_scene_"):-5]
        assert stem in served, stem
        declared = dict(driver)
        declared.update(parse_var_tags(open(path).read()))
        table = {r[0]: r[1:5] for r in oracle.var_table(stem)}
        assert sorted(table) == sorted(declared), stem
        for k, v in declared.items():
            assert tuple(np.float32(x) for x in v) == tuple(np.float32(x) for x in table[k]), (stem, k)


def test_cli_checks_a_scene_source_without_a_gpu(tmp_path, capsys):
    """--scene-source FILE --check compiles a run-time scene for gfx950 on a machine without a GPU."""
    import os

    from sdf_playground_amd import cli

    good = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdf_playground_amd", "scenes", "pendulum.scene.h")
    assert cli.main(["--scene-source", good, "--check"]) == 0
    assert capsys.readouterr().out.strip() == "ok"
    bad = tmp_path / "bad.scene.h"
    bad.write_text(open(good).read().replace("ground_setup(dir)", "ground_setup(dirr)"))
    assert cli.main(["--scene-source", str(bad), "--check"]) == 1
    assert "dirr" in capsys.readouterr().out
