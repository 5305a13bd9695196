// oracle/detmath.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Deterministic fp32 elementary functions for the oracle.
//
// Why this exists: the reference hot path (Engine/shader/pshader_sdf.hlsl and the
// HLSL it includes) calls sin/cos/atan2/pow/exp2/log2 whose D3D implementations are
// not bit-specified.  The forward-difference normal at eps = 1e-4
// (pshader_sdf.hlsl:32,164-177) amplifies 1-ULP differences into visible colour
// differences, so CPU oracle and GPU kernel must evaluate *the same* arithmetic.
// These functions use only IEEE +,-,*,/,fma, rint/floor and bit casts, all of which
// are exactly specified on x86-64 and on gfx950, so both sides produce identical bits.
//
// Polynomial coefficients are the classic Cephes single-precision ones (public
// domain, S. Moshier); range reduction is 3-term Cody-Waite with fma.
// Accuracy (checked in tests/test_oracle_math.py against double libm): <= 2 ulp on
// the ranges the scenes use.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {
namespace dm {

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// x = k*(pi/2) + r, |r| <= pi/4 (for moderate |x|); q = k mod 4 as a float in {0,1,2,3}
inline void reduce_pio2(float x, float &r, float &q)
{
	const float TWO_OVER_PI = 0.636619772367581343f;
	const float P1 = 1.5703125f;                 // pi/2 split, Cody-Waite
	const float P2 = 4.837512969970703125e-4f;
	const float P3 = 7.54978995489188216e-8f;
	float k = rintf(x * TWO_OVER_PI);
	r = fmaf(-k, P1, x);
	r = fmaf(-k, P2, r);
	r = fmaf(-k, P3, r);
	q = k - 4.f * floorf(k * 0.25f);
}

inline float sin_poly(float r)
{
	const float S1 = -1.6666654611e-1f, S2 = 8.3321608736e-3f, S3 = -1.9515295891e-4f;
	float z = r * r;
	float p = fmaf(S3, z, S2);
	p = fmaf(p, z, S1);
	return fmaf(p * z, r, r);
}

inline float cos_poly(float r)
{
	const float C1 = 4.166664568298827e-2f, C2 = -1.388731625493765e-3f, C3 = 2.443315711809948e-5f;
	float z = r * r;
	float p = fmaf(C3, z, C2);
	p = fmaf(p, z, C1);
	return fmaf(p, z * z, fmaf(-0.5f, z, 1.0f));
}

inline float sinf_det(float x)
{
	float r, q;
	reduce_pio2(x, r, q);
	float s = sin_poly(r), c = cos_poly(r);
	float v = (q == 1.f || q == 3.f) ? c : s;
	return (q >= 2.f) ? -v : v;
}

inline float cosf_det(float x)
{
	float r, q;
	reduce_pio2(x, r, q);
	float s = sin_poly(r), c = cos_poly(r);
	float v = (q == 1.f || q == 3.f) ? s : c;
	return (q == 1.f || q == 2.f) ? -v : v;
}

// atan for t >= 0
inline float atan_pos(float t)
{
	const float PIO2 = 1.57079632679489661923f, PIO4 = 0.78539816339744830962f;
	float y = 0.f;
	if (t > 2.414213562373095f) { y = PIO2; t = -(1.0f / t); }
	else if (t > 0.4142135623730950f) { y = PIO4; t = (t - 1.0f) / (t + 1.0f); }
	float z = t * t;
	float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
	p = fmaf(p, z, 1.99777106478e-1f);
	p = fmaf(p, z, -3.33329491539e-1f);
	p = fmaf(p * z, t, t);
	return y + p;
}

// atan2(y, x); atan2(0,0) is defined as 0 here (HLSL leaves it unspecified).
inline float atan2f_det(float y, float x)
{
	const float PI = 3.14159265358979323846f, PIO2 = 1.57079632679489661923f;
	if (x != x || y != y) return x + y;
	float ax = fabsf(x), ay = fabsf(y);
	float a;
	if (ax == 0.f) a = (ay == 0.f) ? 0.f : PIO2;
	else a = atan_pos(ay / ax);
	if (x < 0.f) a = PI - a;
	return (y < 0.f) ? -a : a;
}

inline float exp2f_det(float x)
{
	if (x != x) return x;
	if (x >= 128.f) return u2f(0x7f800000u);
	if (x < -126.f) return 0.f;
	float k = rintf(x);
	float f = x - k;
	float p = 1.535336188319500e-4f;
	p = fmaf(p, f, 1.339887440266574e-3f);
	p = fmaf(p, f, 9.618437357674640e-3f);
	p = fmaf(p, f, 5.550332471162809e-2f);
	p = fmaf(p, f, 2.402264791363012e-1f);
	p = fmaf(p, f, 6.931472028550421e-1f);
	float res = fmaf(p, f, 1.0f);
	int ki = (int)k;
	if (ki > 127) { res = res * 2.f; ki -= 1; }
	return res * u2f((uint32_t)(ki + 127) << 23);
}

inline float log2f_det(float x)
{
	if (x != x) return x;
	if (x < 0.f) return u2f(0x7fc00000u);
	if (x == 0.f) return u2f(0xff800000u);
	if (x == u2f(0x7f800000u)) return x;
	int e = 0;
	if (x < 1.17549435e-38f) { x = x * 16777216.f; e = -24; }
	uint32_t bits = f2u(x);
	e += (int)((bits >> 23) & 0xffu) - 127;
	float m = u2f((bits & 0x007fffffu) | 0x3f800000u);
	if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
	float f = m - 1.0f;
	float z = f * f;
	float p = 7.0376836292e-2f;
	p = fmaf(p, f, -1.1514610310e-1f);
	p = fmaf(p, f, 1.1676998740e-1f);
	p = fmaf(p, f, -1.2420140846e-1f);
	p = fmaf(p, f, 1.4249322787e-1f);
	p = fmaf(p, f, -1.6668057665e-1f);
	p = fmaf(p, f, 2.0000714765e-1f);
	p = fmaf(p, f, -2.4999993993e-1f);
	p = fmaf(p, f, 3.3333331174e-1f);
	float y = (p * f) * z;
	float t = fmaf(-0.5f, z, y);       // ln(m) = f + t
	const float LOG2EA = 0.44269504088896340735992f; // log2(e) - 1
	float r = t * LOG2EA;
	r = fmaf(f, LOG2EA, r);
	r = r + t;
	r = r + f;
	r = r + (float)e;
	return r;
}

// HLSL pow(x, y) = exp2(y * log2(x))  (SURVEY.md a-T.4)
inline float powf_det(float x, float y)
{
	return exp2f_det(y * log2f_det(x));
}

} // namespace dm
} // namespace orc
