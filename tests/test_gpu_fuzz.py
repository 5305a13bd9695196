"""GPU tier: a short, seeded run of the randomised parity hunt (tools/fuzz_parity.py): random
cameras (also far away, grazing the floor), times, variable values, limits and schedules for all
22 scenes against the oracle, bit for bit.  The long runs (190 000 cases in round 1, no mismatch) are
recorded in profiles/README.md."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_randomised_parity_all_scenes():
    import fuzz_parity

    n, bad = fuzz_parity.run(cases=8, seed=7, size=(64, 48))
    assert n == 8 * 22
    assert not bad, bad[:3]


def test_randomised_parity_with_step_shortcuts_and_any_ray_budget():
    """the same hunt on the pixel schedule with step shortcuts on, eight lights in half of the cases and any ray budget (1-16) and
    queue length (1-8): what the escape rules and the shadow rays delivered from the light loop have to respect"""
    import fuzz_parity

    scenes = ["gems", "lense", "fast_sphere", "fractal", "neon", "basic_transparency", "terrain", "light_shadows"]
    n, bad = fuzz_parity.run(cases=40, seed=11, size=(64, 48), scenes=scenes, shortcut_heavy=True)
    assert n == 40 * len(scenes)
    assert not bad, bad[:3]
