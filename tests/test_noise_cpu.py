"""Simplex noise in 2 and 4 dimensions (Engine/shader/noise.hlsl:124-203, 304-433).

Three implementations meet here: the oracle's literal restatement (oracle/noise.h), the product's scalar corner-by-corner
form (sdf_playground_amd/csrc/sdfr_noise.h, built for the CPU by tests/hostsim) and, below, the published algorithm
(Ashima Arts / Stefan Gustavson, webgl-noise, MIT -- the text the reference vendors) written once more in float64 numpy.
Oracle and product must agree bit for bit; both must agree with the float64 version to float32 accuracy; and the analytic
properties of the construction must hold: zero at the lattice points, range, continuity, permute() a permutation of Z/289.
"""
import ctypes

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import hostsim


def _bulk(fn, what, pts):
    pts = np.ascontiguousarray(pts, np.float32)
    n = pts.shape[0]
    out = np.zeros((n, 4) if what == 5 else (n,), np.float32)
    fn(what, pts.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(n))
    return out


def oracle_noise(what, pts):
    return _bulk(po.lib().orc_noise, what, pts)


def product_noise(what, pts):
    return _bulk(hostsim.lib().hostsim_noise, what, pts)


# ---- the published algorithm in float64 -------------------------------------------------------------------------------
def _mod289(x):
    return x - np.floor(x / 289.0) * 289.0


def _permute(x):
    return _mod289((x * 34.0 + 1.0) * x)


def snoise2_f64(v):
    v = np.asarray(v, np.float64)
    C = (0.211324865405187, 0.366025403784439, -0.577350269189626, 0.024390243902439)
    i = np.floor(v + (v[:, :1] + v[:, 1:2]) * C[1])
    x0 = v - i + (i[:, :1] + i[:, 1:2]) * C[0]
    gt = x0[:, 0] > x0[:, 1]
    i1 = np.stack([np.where(gt, 1.0, 0.0), np.where(gt, 0.0, 1.0)], 1)
    x1 = x0 + C[0] - i1
    x2 = x0 + C[2]
    i = _mod289(i)
    p = _permute(_permute(i[:, 1:2] + np.stack([np.zeros(len(v)), i1[:, 1], np.ones(len(v))], 1)) + i[:, :1] + np.stack([np.zeros(len(v)), i1[:, 0], np.ones(len(v))], 1))
    m = np.maximum(0.5 - np.stack([(x0 ** 2).sum(1), (x1 ** 2).sum(1), (x2 ** 2).sum(1)], 1), 0.0)
    m = m ** 4
    x = 2.0 * np.modf(p * C[3])[0] - 1.0
    h = np.abs(x) - 0.5
    a0 = x - np.floor(x + 0.5)
    m = m * (1.79284291400159 - 0.85373472095314 * (a0 * a0 + h * h))
    g = np.stack([a0[:, 0] * x0[:, 0] + h[:, 0] * x0[:, 1], a0[:, 1] * x1[:, 0] + h[:, 1] * x1[:, 1], a0[:, 2] * x2[:, 0] + h[:, 2] * x2[:, 1]], 1)
    return 130.0 * (m * g).sum(1)


def _grad4_f32(j):
    """grad4 in numpy float32, operation by operation.  Which of the 7 x 7 x 6 gradients an index picks is decided by
    floor(frac(j / 7) * 7) and its like: whole numbers up to rounding, so the choice belongs to the float32 arithmetic
    (every j sits on such an edge for the z component) -- float64 picks other, equally valid gradients.  Hence float32
    here, as the shader computes it, and float64 for everything that is continuous."""
    f = np.float32
    j = np.asarray(j, f)
    ip = (f(0.003401360544217687075), f(0.020408163265306122449), f(0.142857142857142857143))
    cols = []
    for k in range(3):
        t = j * ip[k]
        t = t - np.floor(t)
        cols.append(np.floor(t * f(7.0)) * ip[2] - f(1.0))
    p = np.stack(cols, -1)
    a = np.abs(p)
    # dot(abs(p.xyz), ones.xyz) is a fused chain in this library's contract; the products by 1 are exact, so only the two
    # additions round, as they do here
    w = f(1.5) - ((a[..., 0] + a[..., 1]) + a[..., 2])
    p = p - np.sign(p) * (w < 0)[..., None].astype(f)
    return np.concatenate([p, w[..., None]], -1)


def snoise4_f64(v):
    v = np.asarray(v, np.float64)
    G4 = 0.138196601125011
    F4 = 0.309016994374947451
    i = np.floor(v + v.sum(1, keepdims=True) * F4)
    x0 = v - i + i.sum(1, keepdims=True) * G4
    # rank of each component: how many of the others it is >= to (ties: the earlier component wins, as step() does)
    rank = np.zeros_like(x0)
    for a in range(4):
        for b in range(a + 1, 4):
            ge = x0[:, a] >= x0[:, b]
            rank[:, a] += ge
            rank[:, b] += ~ge
    i3 = np.clip(rank, 0, 1)
    i2 = np.clip(rank - 1, 0, 1)
    i1 = np.clip(rank - 2, 0, 1)
    xs = [x0, x0 - i1 + G4, x0 - i2 + 2 * G4, x0 - i3 + 3 * G4, x0 - 1.0 + 4 * G4]
    i = _mod289(i)
    total = np.zeros(len(v))
    for xc, off in zip(xs, [np.zeros_like(x0), i1, i2, i3, np.ones_like(x0)]):
        j = _permute(_permute(_permute(_permute(i[:, 3] + off[:, 3]) + i[:, 2] + off[:, 2]) + i[:, 1] + off[:, 1]) + i[:, 0] + off[:, 0])
        g = _grad4_f32(np.rint(j)).astype(np.float64)
        g = g / np.sqrt((g * g).sum(1, keepdims=True))
        m = np.maximum(0.6 - (xc * xc).sum(1), 0.0)
        total += m ** 4 * (g * xc).sum(1)
    return 49.0 * total


def _points(dim, n, seed, scale):
    rng = np.random.default_rng(seed)
    return ((rng.random((n, dim)) * 2 - 1) * scale).astype(np.float32)


@pytest.mark.parametrize("dim", [2, 3, 4])
def test_product_noise_equals_the_oracle_bit_for_bit(dim):
    for seed, scale in ((1, 3.0), (2, 40.0), (3, 700.0), (4, 0.05)):
        pts = _points(dim, 60000, seed, scale)
        a, b = oracle_noise(dim, pts), product_noise(dim, pts)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (dim, scale, int((a != b).sum()))
    # lattice-aligned and axis points (ties in the rank sorting, zeros)
    grid = np.stack(np.meshgrid(*[np.arange(-3, 4) * 0.5] * dim, indexing="ij"), -1).reshape(-1, dim).astype(np.float32)
    a, b = oracle_noise(dim, grid), product_noise(dim, grid)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_grad4_product_equals_oracle_and_lies_on_the_cross_polytope():
    j = np.arange(0, 289, dtype=np.float32)
    arg = np.stack([j, np.full_like(j, 1 / 294), np.full_like(j, 1 / 49), np.full_like(j, 1 / 7)], 1).astype(np.float32)
    a, b = oracle_noise(5, arg), product_noise(5, arg)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # noise.hlsl:124-138: |x| + |y| + |z| + |w| = 1.5 for every gradient before the fold's sign step; after it the
    # gradient stays within the cube and is never zero
    assert (np.abs(a).sum(1) > 0.4).all() and (np.abs(a[:, :3]) <= 1.0 + 1e-6).all()
    # numpy's float32 arithmetic, operation by operation, picks the same gradients to the bit
    assert np.array_equal(_grad4_f32(j).view(np.uint32), a.view(np.uint32))
    # 7 x 7 x 6 slots for 289 indices: nearly all distinct (float32's rounding at the steps merges a few)
    assert len({tuple(r) for r in a.tolist()}) >= 270


def test_permute_is_a_permutation_of_the_289_ring():
    # noise.hlsl:101-120: (34 x^2 + x) mod 289 is a bijection on 0..288 (the published construction)
    vals = sorted(int(round(float(po.kat("permute", float(k))[0]))) for k in range(289))
    assert vals == list(range(289))
    assert float(po.kat("mod289", 289.0)[0]) == 0.0 and float(po.kat("mod289", -1.0)[0]) == 288.0


@pytest.mark.parametrize("dim,ref", [(2, snoise2_f64), (4, snoise4_f64)])
def test_oracle_noise_follows_the_published_algorithm(dim, ref):
    for seed, scale in ((11, 3.0), (12, 30.0)):
        pts = _points(dim, 20000, seed, scale)
        a = oracle_noise(dim, pts).astype(np.float64)
        b = ref(pts.astype(np.float64))
        # float32 against float64: the permutation indices are exact whole numbers in both, so the only differences are
        # roundings -- except for points within an ulp of a simplex boundary, where the corner sets differ
        close = np.abs(a - b) < 2e-4
        assert close.mean() > 0.999, (dim, scale, float(np.abs(a - b).max()))
        assert np.abs(a - b)[close].max() < 2e-4
        assert np.abs(a).max() <= 1.1  # the published scale factors (130, 49) overshoot 1 by a few per cent in 4-D


def test_noise2_vanishes_on_its_lattice_and_is_continuous():
    # a lattice point of the 2-D simplex grid: v = i - (i.x + i.y) * G2 for whole i; every corner's offset is either 0
    # (its own) or at least sqrt(2/3) > sqrt(0.5) away, so all three kernels vanish
    G2 = 0.211324865405187
    i = np.stack(np.meshgrid(np.arange(-6, 7), np.arange(-6, 7), indexing="ij"), -1).reshape(-1, 2).astype(np.float64)
    v = (i - i.sum(1, keepdims=True) * G2).astype(np.float32)
    assert np.abs(oracle_noise(2, v)).max() < 2e-5
    pts = _points(2, 5000, 5, 4.0)
    d = np.abs(oracle_noise(2, pts + np.float32(1e-4)) - oracle_noise(2, pts))
    assert d.max() < 5e-3
    pts4 = _points(4, 5000, 6, 4.0)
    d4 = np.abs(oracle_noise(4, pts4 + np.float32(1e-4)) - oracle_noise(4, pts4))
    assert np.percentile(d4, 99.5) < 5e-3  # the 4-D kernel radius 0.6 leaves small steps at simplex faces (a known trait)
