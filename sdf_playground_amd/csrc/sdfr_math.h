// sdfr_math.h -- vector types, HLSL-semantics intrinsics and deterministic elementary
// functions for the gfx950 kernels.
//
// Arithmetic contract (DESIGN.md "Arithmetic contract"): every +,-,*,/ and sqrt is one
// IEEE binary32 operation in source order (compile with -ffp-contract=off; hipcc's
// default correctly rounded fp32 divide/sqrt is relied upon), fused multiply-add only
// where written as fma() -- dot/length/lerp/mad/reflect/refract and the polynomial
// kernels below.  v_min_f32/v_max_f32 are used as-is (IEEE minNum/maxNum, -0 < +0).
//
// The header also compiles for the host (g++ or hipcc host pass): the C-ABI uses it to
// pre-compute per-frame scene constants with the very same functions, and
// tests/hostsim builds the per-pixel code on the CPU to bit-compare it with the oracle
// where no GPU is available.  The product never renders on the CPU.
#pragma once
#if !defined(__HIPCC_RTC__) // hiprtc (run-time scene compilation) brings its own runtime declarations
#include <stdint.h>
#include <math.h>
#include <string.h>
#endif

#if defined(__HIPCC_RTC__)
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned long size_t;
#define SDF_HD __host__ __device__ __forceinline__
#elif defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SDF_HD __host__ __device__ __forceinline__
#else
#define SDF_HD inline
#endif

namespace sdfr {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

SDF_HD vec2 V2(float x, float y) { vec2 r; r.x = x; r.y = y; return r; }
SDF_HD vec3 V3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
SDF_HD vec3 V3s(float s) { return V3(s, s, s); }
SDF_HD vec4 V4(float x, float y, float z, float w) { vec4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

SDF_HD uint32_t f32_bits(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __float_as_uint(f);
#else
	uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
SDF_HD float bits_f32(uint32_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __uint_as_float(u);
#else
	float f; memcpy(&f, &u, 4); return f;
#endif
}

// ---- scalar helpers ------------------------------------------------------------------
SDF_HD float fma1(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

SDF_HD float min1(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_fminf(a, b);
#else
	if (a != a) return b;
	if (b != b) return a;
	if (a < b) return a;
	if (b < a) return b;
	return (f32_bits(a) >> 31) ? a : b;
#endif
}
SDF_HD float max1(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_fmaxf(a, b);
#else
	if (a != a) return b;
	if (b != b) return a;
	if (a > b) return a;
	if (b > a) return b;
	return (f32_bits(a) >> 31) ? b : a;
#endif
}
SDF_HD float abs1(float a) { return __builtin_fabsf(a); }
SDF_HD float floor1(float a) { return __builtin_floorf(a); }
SDF_HD float trunc1(float a) { return __builtin_truncf(a); }
SDF_HD float rne1(float a) { return __builtin_rintf(a); }      // HLSL round(): half to even
// IEEE-754 correctly rounded square root.
// Device: hipcc's generic lowering costs ~18 VALU instructions (input scaling, v_sqrt_f32,
// integer next-up/next-down probes, class fix-ups) and sqrt is ~40 % of a scene evaluation,
// so the kernels use Markstein's sequence on the hardware reciprocal-sqrt seed instead:
//     y = rsq(a + 2^-126); g = a*y; h = y/2; g' = g + (a - g*g)*h   (both corrections fused)
// (the 2^-126 bias keeps rsq finite at a = 0 and is below half an ulp of every a >= 2^-96)
// The correctly rounded result is unique, so this is not an approximation: for a = +0 and
// every a in [2^-96, FLT_MAX] the result is bit-identical to sqrtf -- checked exhaustively
// over all 2^32 inputs on gfx950 (sdfr_selftest_math, tests/test_gpu_math.py; the sequence
// is in fact exact down to 2^-102).  Outside that set (negative, -0, denormal/tiny, +inf)
// it is NOT valid; every call site passes a sum of squares or a guarded discriminant of
// scene-scale quantities, and the oracle's census build counts out-of-domain arguments over
// the test and bench workloads (zero).  NaN propagates.
//
// SDFR_SAFE_MATH (what scenes compiled at run time get unless their text asks for the fast forms,
// sdfr_jit.cpp): sqrt1 / rcp1 / div_c are the plain IEEE operations, valid for every input.  The
// built-in scenes are covered by the domain census and the fuzz runs; a user's scene text and
// variable ranges are not, and outside the domains the fast forms silently differ from IEEE.
SDF_HD float sqrt1(float a)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDFR_SAFE_MATH)
	const float y = __builtin_amdgcn_rsqf(a + 0x1p-126f);
	const float g = a * y;
	const float h = 0.5f * y;
	return __builtin_fmaf(__builtin_fmaf(-g, g, a), h, g);
#else
	return __builtin_sqrtf(a);
#endif
}
// the generic lowering, valid for every input (reference for the self-test)
SDF_HD float sqrt_ieee(float a) { return __builtin_sqrtf(a); }
// 1 / a as v_rcp_f32 plus one Newton step: y = rcp(a); y' = fma(fma(-a, y, 1), y, y) -- 3
// instructions (+ a class test and a select for a = 0 or inf, which keep rcp's own result) instead
// of the ~10 of the IEEE divide.  Exhaustively compared on gfx950 with 1.0f / a: bit-identical
// for a = +-0, +-inf and EVERY 2^-100 <= |a| <= 2^100 (sdfr_selftest_math what = 4).
SDF_HD float rcp1(float a)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDFR_SAFE_MATH)
	const float y = __builtin_amdgcn_rcpf(a);
	const float r = __builtin_fmaf(__builtin_fmaf(-a, y, 1.0f), y, y);
	// +-0 (0x60) and +-inf (0x204): rcp is exact there.  __builtin_amdgcn_class is the DOUBLE class test (the float goes
	// through v_cvt_f64_f32): on purpose.  Measured on MI355X (round 2): with the float test (classf: two instructions
	// and the compiler's s_nops fewer per call) labyrinth 4K takes 1.40 instead of 1.38 ms and distortion 5.15 instead of
	// 4.45 ms; with a select-free form (min(e, 2^-20) instead of test + select: four instructions in all) 1.42 and 4.84.
	// Same bits all three (sdfr_selftest_math what = 4), same registers, same spills, no instruction-cache misses
	// (SQC_ICACHE_MISSES 6e3 of 2.7e8 requests), no effect of the distance between v_rcp and its consumer in a
	// micro-benchmark (tools/ubench/select_ubench.hip); under rocprofv3 the shorter form issues 0.4 % fewer VALU instructions
	// and spends 2.4 % more busy cycles (SQ_WAIT_INST_ANY + 7 %).  The cause is not known, the numbers are.
	return __builtin_amdgcn_class(a, 0x260 | 0x204) ? y : r; // +-0 (0x60) and +-inf (0x204): rcp is exact there
#else
	return 1.0f / a;
#endif
}
SDF_HD float rsqrt1(float a) { return rcp1(sqrt1(a)); }

// a / c for a constant c, rc = 1.0f / c folded at compile time:
//     q = a*rc; t = fma(c, q, -a); q' = fma(-t, rc, q)       (= q + (a - c*q)*rc; written so
//     that a = -0 gives -0: the residual of a zero numerator must be -0, not +0)
// instead of the ~11-instruction IEEE divide.  Bit-identical to a / c whenever no
// intermediate under- or overflows: exhaustively checked for the constants the scenes use
// (sdfr_selftest_math) over a = +-0 and 2^-100 <= |a| <= 2^110.  Only for verified constants.
SDF_HD float div_c(float a, float c, float rc)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDFR_SAFE_MATH)
	const float q = a * rc;
	return __builtin_fmaf(-__builtin_fmaf(c, q, -a), rc, q);
#else
	(void)rc;
	return a / c;
#endif
}
SDF_HD float sat1(float a) { return min1(max1(a, 0.f), 1.f); }
SDF_HD float clamp1(float a, float lo, float hi) { return min1(max1(a, lo), hi); }
SDF_HD float step1(float edge, float x) { return x >= edge ? 1.f : 0.f; }
SDF_HD float sign1(float a) { return a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f); }
SDF_HD float frac1(float a) { return a - floor1(a); }
SDF_HD float lerp1(float a, float b, float t) { return fma1(t, b - a, a); }
// fmod as the HLSL compiler expands it: q = a/b; f = frac(|q|); (q >= -q ? f : -f) * b
SDF_HD float fmod1(float a, float b)
{
	float q = a / b;
	float f = frac1(abs1(q));
	return (q >= -q ? f : -f) * b;
}
// fmod1 for a verified constant divisor
SDF_HD float fmod_c(float a, float b, float rb)
{
	float q = div_c(a, b, rb);
	float f = frac1(abs1(q));
	return (q >= -q ? f : -f) * b;
}
SDF_HD float modf1(float a, float *ip) { *ip = trunc1(a); return a - *ip; }
// float -> int conversion with D3D ftoi behaviour (truncate, saturate, NaN -> 0)
SDF_HD int ftoi1(float a)
{
	if (a != a) return 0;
	if (a >= 2147483648.f) return 2147483647;
	if (a <= -2147483648.f) return (int)0x80000000;
	return (int)a;
}

// ---- deterministic sin / cos / atan2 / exp2 / log2 / pow -------------------------------
// Cephes single-precision minimax polynomials, 3-term Cody-Waite reduction with fma.
struct SinCosArg { float r, q; };
SDF_HD SinCosArg sincos_reduce(float x)
{
	float k = rne1(x * 0.636619772367581343f);
	float r = fma1(-k, 1.5703125f, x);
	r = fma1(-k, 4.837512969970703125e-4f, r);
	r = fma1(-k, 7.54978995489188216e-8f, r);
	SinCosArg a;
	a.r = r;
	a.q = k - 4.f * floor1(k * 0.25f);
	return a;
}
SDF_HD float sin_kernel(float r)
{
	float z = r * r;
	float p = fma1(-1.9515295891e-4f, z, 8.3321608736e-3f);
	p = fma1(p, z, -1.6666654611e-1f);
	return fma1(p * z, r, r);
}
SDF_HD float cos_kernel(float r)
{
	float z = r * r;
	float p = fma1(2.443315711809948e-5f, z, -1.388731625493765e-3f);
	p = fma1(p, z, 4.166664568298827e-2f);
	return fma1(p, z * z, fma1(-0.5f, z, 1.0f));
}
SDF_HD float sin1(float x)
{
	SinCosArg a = sincos_reduce(x);
	float s = sin_kernel(a.r), c = cos_kernel(a.r);
	float v = (a.q == 1.f || a.q == 3.f) ? c : s;
	return (a.q >= 2.f) ? -v : v;
}
SDF_HD float cos1(float x)
{
	SinCosArg a = sincos_reduce(x);
	float s = sin_kernel(a.r), c = cos_kernel(a.r);
	float v = (a.q == 1.f || a.q == 3.f) ? s : c;
	return (a.q == 1.f || a.q == 2.f) ? -v : v;
}
// both at once (one range reduction): returns (sin, cos), bit-identical to sin1/cos1
SDF_HD vec2 sincos1(float x)
{
	SinCosArg a = sincos_reduce(x);
	float s = sin_kernel(a.r), c = cos_kernel(a.r);
	bool odd = (a.q == 1.f || a.q == 3.f);
	float vs = odd ? c : s;
	float vc = odd ? s : c;
	return V2((a.q >= 2.f) ? -vs : vs, (a.q == 1.f || a.q == 2.f) ? -vc : vc);
}

// sincos1 for |x| <= 0.75 (< pi/4): there the range reduction finds quadrant 0 and leaves x as it is
// (k = rne(x * 2/pi) = +-0, r = fma(-k, c, x) = x exactly), so the two polynomial kernels alone give the
// bits of sincos1 -- 11 instructions instead of ~30 (no rounding, no quadrant selects).  NaN gives NaN.
SDF_HD vec2 sincos1_small(float x) { return V2(sin_kernel(x), cos_kernel(x)); }

SDF_HD float atan_nonneg(float t)
{
	float y = 0.f;
	if (t > 2.414213562373095f) { y = 1.57079632679489661923f; t = -rcp1(t); }
	else if (t > 0.4142135623730950f) { y = 0.78539816339744830962f; t = (t - 1.0f) / (t + 1.0f); }
	float z = t * t;
	float p = fma1(8.05374449538e-2f, z, -1.38776856032e-1f);
	p = fma1(p, z, 1.99777106478e-1f);
	p = fma1(p, z, -3.33329491539e-1f);
	p = fma1(p * z, t, t);
	return y + p;
}
SDF_HD float atan21(float y, float x)
{
	if (x != x || y != y) return x + y;
	float ax = abs1(x), ay = abs1(y);
	float a;
	if (ax == 0.f) a = (ay == 0.f) ? 0.f : 1.57079632679489661923f;
	else a = atan_nonneg(ay / ax);
	if (x < 0.f) a = 3.14159265358979323846f - a;
	return (y < 0.f) ? -a : a;
}

SDF_HD float atan1(float x) { return atan21(x, 1.f); }

SDF_HD float exp21(float x)
{
	if (x != x) return x;
	if (x >= 128.f) return bits_f32(0x7f800000u);
	if (x < -126.f) return 0.f;
	float k = rne1(x);
	float f = x - k;
	float p = 1.535336188319500e-4f;
	p = fma1(p, f, 1.339887440266574e-3f);
	p = fma1(p, f, 9.618437357674640e-3f);
	p = fma1(p, f, 5.550332471162809e-2f);
	p = fma1(p, f, 2.402264791363012e-1f);
	p = fma1(p, f, 6.931472028550421e-1f);
	float res = fma1(p, f, 1.0f);
	int ki = (int)k;
	if (ki > 127) { res = res * 2.f; ki -= 1; }
	return res * bits_f32((uint32_t)(ki + 127) << 23);
}
SDF_HD float log21(float x)
{
	if (x != x) return x;
	if (x < 0.f) return bits_f32(0x7fc00000u);
	if (x == 0.f) return bits_f32(0xff800000u);
	if (x == bits_f32(0x7f800000u)) return x;
	int e = 0;
	if (x < 1.17549435e-38f) { x = x * 16777216.f; e = -24; }
	uint32_t b = f32_bits(x);
	e += (int)((b >> 23) & 0xffu) - 127;
	float m = bits_f32((b & 0x007fffffu) | 0x3f800000u);
	if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
	float f = m - 1.0f;
	float z = f * f;
	float p = 7.0376836292e-2f;
	p = fma1(p, f, -1.1514610310e-1f);
	p = fma1(p, f, 1.1676998740e-1f);
	p = fma1(p, f, -1.2420140846e-1f);
	p = fma1(p, f, 1.4249322787e-1f);
	p = fma1(p, f, -1.6668057665e-1f);
	p = fma1(p, f, 2.0000714765e-1f);
	p = fma1(p, f, -2.4999993993e-1f);
	p = fma1(p, f, 3.3333331174e-1f);
	float y = (p * f) * z;
	float t = fma1(-0.5f, z, y);
	const float L2EA = 0.44269504088896340735992f;
	float r = t * L2EA;
	r = fma1(f, L2EA, r);
	r = r + t;
	r = r + f;
	r = r + (float)e;
	return r;
}
SDF_HD float pow1(float x, float y) { return exp21(y * log21(x)); }

// ---- vectors -------------------------------------------------------------------------
SDF_HD vec2 operator+(vec2 a, vec2 b) { return V2(a.x + b.x, a.y + b.y); }
SDF_HD vec2 operator-(vec2 a, vec2 b) { return V2(a.x - b.x, a.y - b.y); }
SDF_HD vec2 operator*(vec2 a, vec2 b) { return V2(a.x * b.x, a.y * b.y); }
SDF_HD vec2 operator/(vec2 a, vec2 b) { return V2(a.x / b.x, a.y / b.y); }
SDF_HD vec2 operator+(vec2 a, float b) { return V2(a.x + b, a.y + b); }
SDF_HD vec2 operator-(vec2 a, float b) { return V2(a.x - b, a.y - b); }
SDF_HD vec2 operator*(vec2 a, float b) { return V2(a.x * b, a.y * b); }
SDF_HD vec2 operator/(vec2 a, float b) { return V2(a.x / b, a.y / b); }
SDF_HD vec2 operator-(float a, vec2 b) { return V2(a - b.x, a - b.y); }
SDF_HD vec2 operator*(float a, vec2 b) { return V2(a * b.x, a * b.y); }
SDF_HD vec2 operator-(vec2 a) { return V2(-a.x, -a.y); }

SDF_HD vec3 operator+(vec3 a, vec3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
SDF_HD vec3 operator-(vec3 a, vec3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
SDF_HD vec3 operator*(vec3 a, vec3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
SDF_HD vec3 operator/(vec3 a, vec3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
SDF_HD vec3 operator+(vec3 a, float b) { return V3(a.x + b, a.y + b, a.z + b); }
SDF_HD vec3 operator-(vec3 a, float b) { return V3(a.x - b, a.y - b, a.z - b); }
SDF_HD vec3 operator*(vec3 a, float b) { return V3(a.x * b, a.y * b, a.z * b); }
SDF_HD vec3 operator/(vec3 a, float b) { return V3(a.x / b, a.y / b, a.z / b); }
SDF_HD vec3 operator+(float a, vec3 b) { return V3(a + b.x, a + b.y, a + b.z); }
SDF_HD vec3 operator-(float a, vec3 b) { return V3(a - b.x, a - b.y, a - b.z); }
SDF_HD vec3 operator*(float a, vec3 b) { return V3(a * b.x, a * b.y, a * b.z); }
SDF_HD vec3 operator-(vec3 a) { return V3(-a.x, -a.y, -a.z); }

SDF_HD vec4 operator+(vec4 a, vec4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
SDF_HD vec4 operator-(vec4 a, vec4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
SDF_HD vec4 operator*(vec4 a, vec4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
SDF_HD vec4 operator+(vec4 a, float b) { return V4(a.x + b, a.y + b, a.z + b, a.w + b); }
SDF_HD vec4 operator-(vec4 a, float b) { return V4(a.x - b, a.y - b, a.z - b, a.w - b); }
SDF_HD vec4 operator*(vec4 a, float b) { return V4(a.x * b, a.y * b, a.z * b, a.w * b); }
SDF_HD vec4 operator+(float a, vec4 b) { return V4(a + b.x, a + b.y, a + b.z, a + b.w); }
SDF_HD vec4 operator-(float a, vec4 b) { return V4(a - b.x, a - b.y, a - b.z, a - b.w); }
SDF_HD vec4 operator*(float a, vec4 b) { return V4(a * b.x, a * b.y, a * b.z, a * b.w); }
SDF_HD vec4 operator-(vec4 a) { return V4(-a.x, -a.y, -a.z, -a.w); }

SDF_HD float dot(vec2 a, vec2 b) { return fma1(a.y, b.y, a.x * b.x); }
SDF_HD float dot(vec3 a, vec3 b) { return fma1(a.z, b.z, fma1(a.y, b.y, a.x * b.x)); }
SDF_HD float dot(vec4 a, vec4 b) { return fma1(a.w, b.w, fma1(a.z, b.z, fma1(a.y, b.y, a.x * b.x))); }
SDF_HD float length(vec2 a) { return sqrt1(dot(a, a)); }
SDF_HD float length(vec3 a) { return sqrt1(dot(a, a)); }
SDF_HD vec2 normalize(vec2 a) { return a * rsqrt1(dot(a, a)); }
SDF_HD vec3 normalize(vec3 a) { return a * rsqrt1(dot(a, a)); }
SDF_HD vec3 lerp(vec3 a, vec3 b, float t) { return V3(lerp1(a.x, b.x, t), lerp1(a.y, b.y, t), lerp1(a.z, b.z, t)); }
SDF_HD vec2 lerp(vec2 a, vec2 b, float t) { return V2(lerp1(a.x, b.x, t), lerp1(a.y, b.y, t)); }
// a + b*s, fused per component
SDF_HD vec3 mad(vec3 b, float s, vec3 a) { return V3(fma1(b.x, s, a.x), fma1(b.y, s, a.y), fma1(b.z, s, a.z)); }
SDF_HD vec2 abs(vec2 a) { return V2(abs1(a.x), abs1(a.y)); }
SDF_HD vec3 abs(vec3 a) { return V3(abs1(a.x), abs1(a.y), abs1(a.z)); }
SDF_HD vec4 abs(vec4 a) { return V4(abs1(a.x), abs1(a.y), abs1(a.z), abs1(a.w)); }
SDF_HD vec2 floor(vec2 a) { return V2(floor1(a.x), floor1(a.y)); }
SDF_HD vec3 floor(vec3 a) { return V3(floor1(a.x), floor1(a.y), floor1(a.z)); }
SDF_HD vec4 floor(vec4 a) { return V4(floor1(a.x), floor1(a.y), floor1(a.z), floor1(a.w)); }
SDF_HD vec2 max(vec2 a, float b) { return V2(max1(a.x, b), max1(a.y, b)); }
SDF_HD vec3 max(vec3 a, float b) { return V3(max1(a.x, b), max1(a.y, b), max1(a.z, b)); }
SDF_HD vec4 max(vec4 a, float b) { return V4(max1(a.x, b), max1(a.y, b), max1(a.z, b), max1(a.w, b)); }
SDF_HD vec3 max(vec3 a, vec3 b) { return V3(max1(a.x, b.x), max1(a.y, b.y), max1(a.z, b.z)); }
SDF_HD vec3 min(vec3 a, vec3 b) { return V3(min1(a.x, b.x), min1(a.y, b.y), min1(a.z, b.z)); }
SDF_HD vec3 saturate(vec3 a) { return V3(sat1(a.x), sat1(a.y), sat1(a.z)); }
SDF_HD bool any3(vec3 a) { return a.x != 0.f || a.y != 0.f || a.z != 0.f; }

SDF_HD vec3 reflect(vec3 i, vec3 n)
{
	float k = 2.f * dot(i, n);
	return V3(fma1(-k, n.x, i.x), fma1(-k, n.y, i.y), fma1(-k, n.z, i.z));
}
SDF_HD vec3 refract(vec3 i, vec3 n, float eta)
{
	float d = dot(n, i);
	float k = fma1(-(eta * eta), fma1(-d, d, 1.f), 1.f);
	if (k < 0.f) return V3s(0.f);
	float s = fma1(eta, d, sqrt1(k));
	return V3(fma1(-s, n.x, eta * i.x), fma1(-s, n.y, eta * i.y), fma1(-s, n.z, eta * i.z));
}

} // namespace sdfr
