"""GPU tier: bench.py end to end on a small frame -- the single-rank path and the strips + RCCL
gather + assembly pipeline (two frames in flight) forced onto one rank; rank 0's assembled image
must equal a direct render bit for bit, and stdout must carry exactly one JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", "328", "--height", "205", "--steps", "7", "--warmup", "2",
                        "--no-cpu-baseline", "--verify"] + extra, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_single_rank_line():
    d = _run([])
    assert d["verified"] is True and d["n_gpus"] == 1 and d["steps"] == 7 and d["unit"] == "Mrays/s"
    assert d["value"] > 0 and d["roofline"]["bound"] == "valu" and d["roofline"]["kernel_ms"] > 0
    assert d["config"]["parallelism"] == "single"


def test_bench_strip_pipeline_assembles_the_same_image():
    d = _run(["--force-distributed"])
    assert d["verified"] is True
    assert d["config"]["parallelism"] == "strips1" and d["scaling"] == "strong"
    cal = d["config"]["strip_calibration"]
    assert cal["t_full_ms"] > 0 and cal["t_gather_ms"] > 0 and 0 <= cal["private_strips_of_16"] < 16


def test_bench_strip_pipeline_with_private_strips():
    """rank 0 keeps 5 of every 16 strips for itself (rendered straight into the image, never gathered)"""
    d = _run(["--force-distributed", "--private-strips", "5"])
    assert d["verified"] is True and d["config"]["strip_calibration"]["private_strips_of_16"] == 5
    # same frames, same rays as the plain pipeline
    assert abs(d["config"]["rays_per_pixel"] - _run(["--force-distributed", "--private-strips", "0"])["config"]["rays_per_pixel"]) < 1e-12
