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
