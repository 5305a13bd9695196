"""CPU tier: the oracle's restatement of HDR::process (oracle/postprocess.h) -- known answers
for bloom.hlsl / pshader_hdr.hlsl semantics and the fp16 / unorm8 conversions."""
import math

import numpy as np

COEFFS = np.array([0.070771, 0.069674, 0.066483, 0.061487, 0.055116, 0.047886, 0.040324, 0.032912, 0.026035,
                   0.019962, 0.014834, 0.010685, 0.007459, 0.005047, 0.003310, 0.002104, 0.001296], np.float32)


def test_half_conversions_match_numpy(oracle):
    h = np.arange(65536, dtype=np.uint16)
    f = oracle.half_to_float(h)
    ref = h.view(np.float16).astype(np.float32)
    ok = ~np.isnan(ref)
    assert np.array_equal(f.view(np.uint32)[ok], ref.view(np.uint32)[ok]) and np.all(np.isnan(f[~ok]))
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(100000) * rng.choice([1e-8, 1e-5, 1e-3, 1, 100, 60000], 100000)).astype(np.float32)
    x = np.concatenate([x, f[ok], np.array([65504, 65519.99, 65520, 6e-8, 2.98e-8, 2.9802322e-8, 2.9802326e-8, 0, -0.0, np.inf], np.float32)])
    with np.errstate(over="ignore"):
        assert np.array_equal(oracle.float_to_half(x), x.astype(np.float16).view(np.uint16))


def test_impulse_response_is_the_separable_kernel(oracle):
    img = np.zeros((80, 100, 4), np.float16)
    img[40, 50] = [4, 4, 4, 1]  # brightness 4 -> factor 1
    b1, b2, ldr = oracle.postprocess(img)
    row = b1[40, :, 0].astype(np.float32)
    for i in range(-16, 17):
        want = np.float16(np.float32(np.float32(4.0) * COEFFS[abs(i)]) * np.float32(2.0))
        assert row[50 + 2 * i] == np.float32(want)
        if i < 16:
            assert row[50 + 2 * i + 1] == 0  # stride 2: odd offsets untouched
    assert np.count_nonzero(b1[:, :, 0]) == 33 and np.count_nonzero(b2[:, :, 0]) == 33 * 33
    assert b2[40 + 2 * 3, 50 - 2 * 5, 1] == np.float16(np.float32(np.float32(b1[40, 50 - 10, 1]) * COEFFS[3]) * np.float32(2.0))


def test_bright_pass_threshold_and_alpha_scaling(oracle):
    img = np.zeros((8, 70, 4), np.float16)
    img[4, 35] = [0.7, 0.7, 0.7, 1]  # brightness 0.7 < 0.75 -> no bloom
    b1, _, _ = oracle.postprocess(img)
    assert not b1.any()
    img[4, 35] = [0.875, 0.875, 0.875, 1]  # factor = (0.875 - 0.75) * 4 = 0.5 (brightness = 0.875 * 1.0)
    b1, _, _ = oracle.postprocess(img)
    f = np.float32(np.float16(0.875))
    br = np.float32(math.fma(float(f), float(np.float32(0.0722)), float(np.float32(math.fma(float(f), float(np.float32(0.7152)), float(f * np.float32(0.2126))))))) if hasattr(math, "fma") else None
    assert 0.0 < float(b1[4, 35, 0]) < 2 * 0.070771 * 0.875
    assert float(b1[4, 35, 3]) > 0  # `col *= factor` scales alpha too (bloom.hlsl:23)


def test_out_of_range_texels_read_zero(oracle):
    img = np.zeros((5, 5, 4), np.float16)
    img[:, :] = [2, 2, 2, 1]
    b1, b2, _ = oracle.postprocess(img)
    # at x = 0 only taps i = 0, 1, 2 are inside (x + 2i in {0, 2, 4})
    want = np.float32(0)
    for i in range(-16, 17):
        if 0 <= 0 + 2 * i < 5:
            want = np.float32(want + np.float32(np.float32(2.0) * COEFFS[abs(i)]))
    assert b1[2, 0, 0] == np.float16(want * np.float32(2.0))


def test_tone_map_and_alpha_flag(oracle):
    img = np.zeros((4, 4, 4), np.float16)
    img[1, 1] = [0.5, 0.25, 0.0, 1.0]   # tone-mapped
    img[2, 2] = [0.5, 0.25, 0.0, 0.0]   # alpha 0: passed through (pshader_hdr.hlsl:25)
    _, _, ldr = oracle.postprocess(img)
    for c, v in enumerate((0.5, 0.25, 0.0)):
        assert abs(int(ldr[1, 1, c]) - int((1 - math.exp(-v)) * 255 + 0.5)) <= 1
        assert ldr[2, 2, c] == int(v * 255 + 0.5)
    assert ldr[2, 2, 3] == 0 and ldr[0, 0].tolist() == [0, 0, 0, 0]
    # unorm8 conversion saturates and rounds half up
    img[3, 3] = [8.0, -1.0, 0.50196, 0.0]
    _, _, ldr = oracle.postprocess(img)
    assert ldr[3, 3, 0] == 255 and ldr[3, 3, 1] == 0 and ldr[3, 3, 2] == 128
