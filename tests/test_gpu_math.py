"""GPU tier: the kernels' fast exact arithmetic.  sqrt1 (Markstein sequence on v_rsq_f32) and
div_c (reciprocal + fused correction for scene constants) must return the SAME correctly
rounded bits as the generic IEEE lowering on their whole stated domain -- checked exhaustively
(every fp32 bit pattern) on the GPU through the C ABI (sdfr_selftest_math)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# every divisor handed to div_c / op_rep_inf_c / op_pipe_c / turbulence3 by the scene code
SCENE_DIVISORS = [
    20.0,                                              # labyrinth cell size
    3.0, 10.0,                                         # lense background cells
    15.0,                                              # turbulence3 normalisation
    float(np.float32(6.28318530717958647)),            # op_rep_angle: angle * count / tau
    float(np.float32(0.001)), float(np.float32(0.05)), float(np.float32(0.025)), float(np.float32(0.01)),   # smooth min / max widths (gems, table, tree)
    float(np.float32(np.float32(np.float32(1.41421356237309504) * np.float32(0.1)) / np.float32(4.0))),  # op_pipe period (labyrinth vase)
]
# tree: the branch generations' scales 1.4^-i, formed like the scene does (repeated fp32 division)
_scale = np.float32(1.0)
for _i in range(8):
    _scale = np.float32(_scale / np.float32(1.4))
    SCENE_DIVISORS.append(float(_scale))
SCENE_DIVISORS += [float(np.float32(2.2)), 11.0]                  # tree: lattice spacing, hop period (10 is in the list already)
SCENE_DIVISORS += [float(3 ** i) for i in range(2, 8)]   # fractal: level scales 9 .. 2187 (3 is in the list already)


@pytest.fixture(scope="module")
def renderer():
    import sdf_playground_amd as sp

    r = sp.SDFRenderer(0)
    yield r
    r.close()


def test_fast_sqrt_is_correctly_rounded_on_its_domain(renderer):
    assert renderer.selftestMath(0) == 0


@pytest.mark.parametrize("c", SCENE_DIVISORS)
def test_constant_division_is_exact(renderer, c):
    assert renderer.selftestMath(1, c) == 0


def test_fast_ground_plane_division_is_exact(renderer):
    """height / denominator with a per-ray reciprocal (sdfr_scenes.h: ground_dist): any
    denominator the fast plane can produce, all heights in range."""
    rng = np.random.default_rng(1)
    cs = list(np.exp(rng.uniform(np.log(1e-20), np.log(2.0), 48)).astype(np.float32)) + [np.float32(1e-20), np.float32(1.0), np.float32(1.0) + np.float32(1e-20)]
    for e in (-60, -20, -1, 0):  # adversarial significands
        for m in (0x7FFFFF, 0x000001, 0x555555, 0x400001):
            cs.append(np.uint32(((127 + e) << 23) | m).view(np.float32))
    for c in cs:
        assert renderer.selftestMath(3, float(c)) == 0, float(c)


def test_fast_reciprocal_is_exact(renderer):
    """rcp1 = v_rcp_f32 + one Newton step: the IEEE 1 / a for +-0, +-inf and every 2^-100 <= |a| <= 2^100."""
    assert renderer.selftestMath(4) == 0
    assert renderer.selftestMath(5) > 0      # negative control: the bare hardware reciprocal is not


def test_selftest_is_not_vacuous(renderer):
    # negative control: the plain reciprocal multiply is NOT the correctly rounded quotient
    assert renderer.selftestMath(2, 3.0) > 0
    assert renderer.selftestMath(2, 20.0) > 0
