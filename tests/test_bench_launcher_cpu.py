"""CPU tier: `python bench.py --gpus N` starts N rank processes by itself (the driver may call it
without a launcher), before anything touches a GPU, and relays rank 0's single line; a launcher's
WORLD_SIZE that contradicts --gpus is refused.  --dry-launch lets the ranks report what they see
over gloo instead of rendering."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)


def test_gpus_2_starts_two_ranks_with_the_right_environment():
    p = _bench(["--gpus", "2", "--dry-launch", "--steps", "3"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout  # ONE line on stdout, whatever the children print
    d = json.loads(lines[0])
    assert d["dry_launch"] is True and d["n_gpus"] == 2 and d["gpus_arg"] == 2
    ranks = sorted(d["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1] and [r["local_rank"] for r in ranks] == [0, 1]
    assert all(r["world_size"] == 2 and r["master_addr"] == "127.0.0.1" and r["ipc_mode_legacy"] == "0" for r in ranks)
    assert len({r["pid"] for r in ranks}) == 2 and os.getpid() not in {r["pid"] for r in ranks}  # fresh processes


def test_single_rank_needs_no_launcher():
    p = _bench(["--dry-launch"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip())
    assert d["n_gpus"] == 1 and d["ranks"][0]["rank"] == 0


def test_world_size_that_contradicts_gpus_is_refused():
    p = _bench(["--gpus", "2", "--dry-launch"], env={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr and not p.stdout.strip()
    p = _bench(["--dry-launch"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode != 0  # --gpus defaults to 1


def test_a_failing_rank_fails_the_parent():
    # an unknown flag makes every child exit non-zero; the parent must not print a line and must fail
    p = _bench(["--gpus", "2", "--dry-launch", "--no-such-flag"])
    assert p.returncode != 0 and not p.stdout.strip()


def c_ref(bench):
    """the reflective variant is config 3 plus ONE labelled extension, and says so"""
    a, b = bench.CONFIGS["3"], bench.CONFIGS["3r"]
    return (b["scene"], b["width"], b["height"], b["camera"]) == (a["scene"], a["width"], a["height"], a["camera"]) and \
        b["limits"] == dict(a["limits"], extension_marble_reflection=0.25) and "LABELLED EXTENSION" in b["workload"] and "extension" in b["metric"]


def test_configurations_are_the_ones_survey_8d_defines():
    import bench

    assert sorted(bench.CONFIGS) == ["1", "2", "3", "3r", "4", "5", "5g"]
    assert c_ref(bench)
    c = bench.CONFIGS
    assert (c["2"]["scene"], c["2"]["width"], c["2"]["height"], c["2"]["limits"]["iter_count"]) == ("cube_sea", 1920, 1080, 128)
    assert (c["3"]["scene"], c["3"]["width"], c["3"]["height"], c["3"]["limits"]) == ("labyrinth", 3840, 2160, {"iter_count": 256})
    assert (c["4"]["scene"], c["4"]["limits"]["iter_count"]) == ("fractal", 512)
    for k in ("5", "5g"):
        assert c[k]["limits"] == {"iter_count": 100, "max_cost_default": 9, "extension_lights": 7}
    # the sweeps are deterministic and differ frame to frame
    for k, cfg in c.items():
        cams = [cfg["camera"](i) for i in range(bench.SWEEP)]
        assert cams == [cfg["camera"](i) for i in range(bench.SWEEP)] and len({cm[1] for cm in cams}) == (1 if k == "1" else bench.SWEEP)
    # configuration 1: the reference's start-up camera, no secondary rays (Application.cpp:214-224; SURVEY.md 8d cfg 1)
    assert (c["1"]["scene"], c["1"]["width"], c["1"]["height"], c["1"]["limits"]) == ("fast_sphere", 256, 256, {"iter_count": 64, "max_cost_default": 2})
    assert c["1"]["camera"](7) == ("lookat", (0.0, 2.0, -3.0), (0.0, 1.0, 0.0), 0.0)
    # the full-size GPU parity test and the bench agree on the cameras (frame 5)
    import tests.test_gpu_fullsize as tf

    by_scene = {cfg[0]: cfg for cfg in tf._configs() if "extension_marble_reflection" not in cfg[3]}
    for k, scene in (("2", "cube_sea"), ("3", "labyrinth"), ("4", "fractal"), ("5", "lense"), ("5g", "gems")):
        kind, eye, tgt, stime = c[k]["camera"](5)
        assert by_scene[scene][4] == (kind, eye, tgt) and by_scene[scene][5] == stime and by_scene[scene][3] == c[k]["limits"]


def test_config_1_is_the_cpu_line_and_needs_no_gpu():
    """BASELINE.json configs[0] (SURVEY.md 8d cfg 1): fast_sphere 256x256, 64 steps, no secondary rays, the CPU scalar
    raymarch in full -- `bench.py --config 1` prints the contract's one JSON line from the oracle alone"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "1", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 0 and d["unit"] == "Mrays/s" and d["value"] > 0 and d["steps"] == 3 and d["config"]["baseline_config"] == "1"
    assert d["config"]["rays_per_pixel"] == 1.0  # max_cost_default 2: no child ray is ever spawned
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and "whole 256x256 frame" in d["cpu_baseline"]["sample"]
