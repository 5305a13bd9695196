"""Pins the CPU oracle: the reference's own known-answer tests for this path
(UnitTest/UnitTest.cpp), the reference's Camera/Math3D compiled here (oracle/_ref), the
camera golden of SURVEY.md 8(c), and analytic known answers for the restated library."""
import math

import numpy as np
import pytest


# ---- reference KATs: UnitTest/UnitTest.cpp:184-229 ---------------------------------
# TestFastSphere1/5/6 agree with the shipped HLSL (sdf_primitives.hlsl:11-45).
@pytest.mark.parametrize("pos,expected", [((-4.0, 0, 0), 3.0), ((0.5, 0, 0), 0.5), ((-0.5, 0, 0), 1.5)])
def test_fast_sphere_reference_kats(oracle, pos, expected):
    d = oracle.kat("sdSphereFast", *pos, 1.0, 0.0, 0.0, 1.0, 1.0)[0]
    assert abs(d - expected) < 1e-3  # tolerance of UnitTest.cpp:55-58


# TestFastSphere2/3/4 test a diverged C++ copy: the shipped HLSL
# returns the 1e10 sentinel on every miss.  Recorded as known divergences.
@pytest.mark.parametrize("pos,dir", [((-4.0, 2, 0), (1.0, 0, 0)), ((0.0, 2, 0), (1.0, 0, 0)), ((0.0, 2, 0), (1.0, 0.1, 0))])
def test_fast_sphere_known_divergences(oracle, pos, dir):
    assert oracle.kat("sdSphereFast", *pos, *dir, 1.0, 1.0)[0] == np.float32(1e10)


def test_fast_sphere_slow_path(oracle):
    # dir.w == 0 -> exact sphere distance (sdf_primitives.hlsl:41-44)
    assert oracle.kat("sdSphereFast", -4.0, 0, 0, 1.0, 0, 0, 0.0, 1.0)[0] == 3.0


# ---- reference KATs: UnitTest/UnitTest.cpp:91-179 (splitString) ---------------------
SPLIT_CASES = [
    ("hello;world", ";", "", ["hello", "world"], [";"]),
    ("hello<>world", "<>", "", ["hello", "world"], ["<>"]),
    ("hello<>world", "<", ">", ["hello", "world"], ["<>"]),
    ("hello<abc>world", "<", ">", ["hello", "world"], ["<abc>"]),
    ("hello<abc>world<def>", "<", ">", ["hello", "world", ""], ["<abc>", "<def>"]),
    ("hello<abc>world<def>huhu", "<", ">", ["hello", "world", "huhu"], ["<abc>", "<def>"]),
    ("hello<abc>wo<rld<def>huhu", "<", ">", ["hello", "wo", "huhu"], ["<abc>", "<rld<def>"]),
]


@pytest.mark.parametrize("s,a,b,parts,seps", SPLIT_CASES)
def test_split_string_reference_kats(oracle, s, a, b, parts, seps):
    p, q = oracle.split_string(s, a, b)
    assert p == parts and q == seps


def test_remove_spaces(oracle):
    assert oracle.remove_spaces("  min ") == "min"
    assert oracle.remove_spaces("\t+10") == "+10"


# ---- VAR_ parser (ShaderUtil.cpp:122-191, README.md:107-110) ------------------------
def test_var_parser_defaults_and_quirks(oracle):
    v = oracle.parse_vars("float a = VAR_foo(); float b = VAR_bar(min = -4, max = 4, step = 0.1); VAR_q(min=1,max=3,start=1.5)")
    assert v["foo"] == (0.0, 2.0, 1.0, np.float32(0.1), 1.0)  # min 0, max 2, start mid, step 5 %
    assert v["bar"][:3] == (-4.0, 4.0, 0.0) and v["bar"][3] == np.float32(0.1)
    assert v["q"][2] == 1.5 and v["q"][4] == 1.5
    # unknown keys are ignored (scenes/sdf_scene_tiling.hlsl:75 uses `steps=`)
    v = oracle.parse_vars("VAR_t(min = 0, max = 10, steps = 1)")
    assert v["t"][3] == np.float32(0.5)
    # a name seen twice keeps the last definition
    v = oracle.parse_vars("VAR_x(min=0,max=1) VAR_x(min=0,max=4)")
    assert v["x"][1] == 4.0


def test_var_table_order_and_defaults(oracle):
    names = [r[0] for r in oracle.var_table("lense")]
    assert names == sorted(names)  # std::map order (ShaderUtil.cpp:257-267)
    assert names == ["debug_nx", "debug_ny", "debug_nz", "debug_scale", "debug_x", "debug_y", "debug_z", "mixing", "show_objects", "xpos", "ypos", "zpos"]
    t = {r[0]: r for r in oracle.var_table("lense")}
    assert t["show_objects"][3] == 1.0 and t["debug_scale"][3] == np.float32(0.2)
    assert t["zpos"][3] == 12.5 and t["mixing"][3] == 0.5
    assert [r[0] for r in oracle.var_table("labyrinth")] == ["debug_nx", "debug_ny", "debug_nz", "debug_scale", "debug_x", "debug_y", "debug_z", "show_objects"]


# ---- camera: golden of SURVEY.md 8(c) + the reference's own code --------------------
FOVY = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)  # Math3D.h:299-303


def test_camera_startup_golden(oracle):
    b = oracle.camera_lookat((0, 2, -3), (0, 1, 0), FOVY, np.float32(1200) / np.float32(800))
    hexes = [float(x).hex() for x in b.ravel()]
    assert hexes[4] == "-0x1.43d1360000000p-2" and hexes[5] == "0x1.e5b9d00000000p-1"
    assert hexes[6] == "0x1.bb67b00000000p-1" and hexes[7] == hexes[8] == "0x0.0p+0"
    assert hexes[10] == "0x1.186f1a0000000p-1" and hexes[11] == "0x1.75e97c0000000p-3"


def test_camera_matches_compiled_reference(oracle):
    ref = oracle.ref_camera_lib()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(7)
    for k in range(200):
        eye = rng.uniform(-8, 8, 3).astype(np.float32)
        tgt = rng.uniform(-8, 8, 3).astype(np.float32)
        fovy = np.float32(rng.uniform(0.3, 2.0))
        aspect = np.float32(rng.uniform(0.5, 2.5))
        roll = np.float32(0.0 if k % 2 == 0 else rng.uniform(-1, 1))
        for is_dir in (0, 1):
            out = np.zeros(12, np.float32)
            ref.ref_camera_basis(eye.ctypes.data, tgt.ctypes.data, is_dir, fovy, aspect, roll, out.ctypes.data)
            mine = (oracle.camera_direction if is_dir else oracle.camera_lookat)(eye, tgt, fovy, aspect, roll)
            assert np.array_equal(mine.ravel().view(np.uint32), out.view(np.uint32))


# ---- analytic known answers --------------------------------------------------------
def pcg_hash(x):  # independent restatement of noise.hlsl:6-11 in Python integers
    state = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def test_hash_uint32_wraparound(oracle):
    for x in [0, 1, 2, 3, 12345, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0x5DF00003]:
        got = int(oracle.kat_u32("hash", x).view(np.uint32)[0])
        assert got == pcg_hash(x)
        hf = oracle.kat_u32("hashf", x)[0]
        assert hf == np.float32(np.float32(pcg_hash(x)) / np.float32(4294967295.0))


def test_hlsl_intrinsics(oracle):
    k = oracle.kat
    # round = half-to-even (a-T.1)
    assert [k("round", v)[0] for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999)] == [0.0, 2.0, 2.0, -0.0, -2.0, 2.0]
    # fmod keeps the sign of the dividend (a-T.2)
    assert abs(k("fmod", 5.5, 2.0)[0] - 1.5) < 1e-6 and abs(k("fmod", -5.5, 2.0)[0] + 1.5) < 1e-6
    assert abs(k("fmod", 0.3, 0.0353553)[0] - math.fmod(0.3, 0.0353553)) < 1e-6
    assert k("step", 0.0, 0.0)[0] == 1.0 and k("step", 0.0, -1e-30)[0] == 0.0
    assert [k("sign", v)[0] for v in (-2.0, 0.0, 3.0)] == [-1.0, 0.0, 1.0]
    assert k("frac", -0.25)[0] == 0.75
    m = k("modf", -2.75)
    assert m[0] == -0.75 and m[1] == -2.0
    # min/max: NaN loses, -0 < +0 (a-T.5)
    assert k("min", float("nan"), 1.0)[0] == 1.0 and k("max", 2.0, float("nan"))[0] == 2.0
    assert math.copysign(1, k("min", 0.0, -0.0)[0]) == -1 and math.copysign(1, k("max", -0.0, 0.0)[0]) == 1
    # pow = exp2(y log2 x) (a-T.4)
    assert k("pow", 0.0, 60.0)[0] == 0.0 and math.isnan(k("pow", -1.0, 2.0)[0])
    assert abs(k("pow", 0.1, 0.25)[0] - 0.1 ** 0.25) < 2e-7


def test_refract_and_reflect(oracle):
    r = oracle.kat("reflect", 1.0, -1.0, 0.0, 0.0, 1.0, 0.0)
    assert np.allclose(r, [1, 1, 0])
    # total internal reflection -> zero vector (Q7)
    i = np.array([math.sin(1.2), -math.cos(1.2), 0.0])
    assert np.all(oracle.kat("refract", *i, 0.0, 1.0, 0.0, 1.4) == 0.0)
    # straight through at normal incidence
    assert np.allclose(oracle.kat("refract", 0.0, -1.0, 0.0, 0.0, 1.0, 0.0, 1 / 1.4), [0, -1, 0], atol=1e-6)
    # Snell
    t = oracle.kat("refract", *i, 0.0, 1.0, 0.0, 1 / 1.4)
    assert abs(abs(t[0]) - math.sin(1.2) / 1.4) < 1e-6


def test_primitives_analytic(oracle):
    k = oracle.kat
    assert k("sdSphere", 3.0, 4.0, 0.0, 1.0)[0] == 4.0
    assert k("sdBox", 3.0, 0.0, 0.0, 1.0, 1.0, 1.0)[0] == 2.0
    assert abs(k("sdBox", 2.0, 2.0, 0.0, 1.0, 1.0, 1.0)[0] - math.sqrt(2)) < 1e-6
    assert k("sdBox", 0.0, 0.0, 0.0, 1.0, 2.0, 3.0)[0] == -1.0
    assert k("sdPlane", 1.0, 2.0, 3.0, 0.0, 1.0, 0.0)[0] == 2.0
    # fast plane: distance ALONG the ray (sdf_primitives.hlsl:59-70)
    d = k("sdPlaneFast", 0.0, 2.0, 0.0, 0.0, -0.5, math.sqrt(0.75), 1.0, 0.0, 1.0, 0.0)[0]
    assert abs(d - 4.0) < 1e-5
    # ray pointing away: saturate(...) = 0 -> plane/1e-20
    assert k("sdPlaneFast", 0.0, 2.0, 0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0)[0] > 1e19
    assert abs(k("sdTorusXY", 3.0, 0.0, 0.0, 2.0, 0.5)[0] - 0.5) < 1e-6
    assert abs(k("sdCappedCylinder", 2.0, 0.0, 0.0, 1.0, 0.5)[0] - 1.5) < 1e-6
    # round cone: on the axis below the big end
    assert abs(k("sdRoundCone", 0.0, 0.0, 0.0, 0.0, 1.1, 0.0, 0.0, 1.6, 0.0, 0.15, 0.1)[0] - (1.1 - 0.15)) < 1e-5
    # guard object: distance along the ray to the cell wall (sdf_primitives.hlsl:118-124)
    assert abs(k("sdLimit2", 0.0, 0.0, 1.0, 0.0, 2.01, 2.01)[0] - 1.005) < 1e-6
    assert abs(k("opRepInf", 7.3, 2.0)[0] - (-0.7)) < 1e-6
    p = k("opRotate", 1.0, 0.0, math.pi / 2)
    assert np.allclose(p, [0, 1], atol=1e-6)
    q = k("opRepAngle", math.cos(1.0), math.sin(1.0), 8.0)
    assert q[2] == 1.0 and abs(math.atan2(q[1], q[0]) - (1.0 - 2 * math.pi / 8)) < 1e-5


def test_noise_properties(oracle):
    rng = np.random.default_rng(3)
    vals = []
    for _ in range(2000):
        p = rng.uniform(-50, 50, 3)
        vals.append(oracle.kat("snoise3", *p)[0])
    vals = np.array(vals)
    assert np.all(np.isfinite(vals)) and vals.min() >= -1.0 and vals.max() <= 1.0
    assert vals.std() > 0.2  # not degenerate
    # continuity: simplex noise is smooth
    a = oracle.kat("snoise3", 1.2345, 2.3456, 3.4567)[0]
    b = oracle.kat("snoise3", 1.2345 + 1e-4, 2.3456, 3.4567)[0]
    assert abs(a - b) < 1e-2
    t = oracle.kat("turbulence", 0.3, 0.7, -1.9)[0]
    s = [oracle.kat("snoise3", 0.3 * f, 0.7 * f, -1.9 * f)[0] for f in (1, 2, 4, 8)]
    assert abs(t - (s[0] + s[1] / 2 + s[2] / 4 + s[3] / 8) * 8 / 15) < 1e-6


def test_checker_parity(oracle):
    # sdf_common.hlsl:30-41: tiles alternate 0.1 / 0.8 grey
    c00 = oracle.kat("tile_color", 0.5, 0.5)
    c10 = oracle.kat("tile_color", 1.5, 0.5)
    c11 = oracle.kat("tile_color", 1.5, 1.5)
    assert c00[0] != c10[0] and c00[0] == c11[0]
    assert {float(c00[0]), float(c10[0])} == {float(np.float32(0.1)), float(np.float32(0.8))}
    assert abs(c00[3] - 0.5) < 1e-7  # distance to the tile border at the centre


def test_detmath_accuracy(oracle):
    rng = np.random.default_rng(11)
    xs = rng.uniform(-2000, 2000, 4000).astype(np.float32)
    s = np.array([oracle.kat("sin", x)[0] for x in xs])
    c = np.array([oracle.kat("cos", x)[0] for x in xs])
    assert np.max(np.abs(s - np.sin(xs.astype(np.float64)))) < 3e-7
    assert np.max(np.abs(c - np.cos(xs.astype(np.float64)))) < 3e-7
    for y, x in rng.uniform(-5, 5, (500, 2)):
        assert abs(oracle.kat("atan2", y, x)[0] - math.atan2(np.float32(y), np.float32(x))) < 1e-6
    for x in rng.uniform(-100, 100, 500):
        r = 2.0 ** float(np.float32(x))
        assert abs(oracle.kat("exp2", x)[0] - r) <= 2e-7 * r
    for x in np.exp(rng.uniform(-60, 60, 500)):
        r = math.log2(float(np.float32(x)))
        assert abs(oracle.kat("log2", x)[0] - r) <= 3e-7 * max(1.0, abs(r))
