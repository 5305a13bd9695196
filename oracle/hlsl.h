// oracle/hlsl.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// HLSL value types and intrinsics restated for a scalar CPU build, with the HLSL
// semantics listed in SURVEY.md section 8 a-T (round = half-to-even, fmod = FXC
// expansion, min/max = IEEE minNum/maxNum, pow = exp2(y*log2 x), step/sign/lerp/...).
//
// Arithmetic contract shared (by specification, not by code) with the HIP kernels:
//   * every +,-,*,/ and sqrt is one IEEE-754 binary32 operation, evaluated in source
//     order; no contraction (-ffp-contract=off);
//   * fused multiply-add is used ONLY inside the named helpers below: dot2/3/4 (and
//     therefore length/normalize), lerp, mad, reflect, refract, and inside detmath.h;
//   * rsqrt(x) = 1/sqrt(x) (two correctly rounded operations);
//   * min/max of (+0,-0) order the zeros as -0 < +0 (what v_min_f32/v_max_f32 do).
//
// `real` is float in the normal build.  With -DORACLE_CENSUS it is a wrapper that
// counts arithmetic operations (+ - * / sqrt rsqrt transcendental
// min max compare select = 1, fma = 2, abs/neg = 0) for the roofline numerator.
#pragma once
#include "detmath.h"
#include <cstdio>
#include <cstdlib>

namespace orc {

#ifdef ORACLE_CENSUS
struct Census
{
	unsigned long long flops, transc;
	// arguments outside the domain on which the GPU kernels' fast exact sequences are valid
	// (sdf_playground_amd/csrc/sdfr_math.h: sqrt1, div_c).  "near": negative, -0 or tiny
	// arguments -- must stay 0 on every workload.  "far": overflowed (+inf / NaN / > 2^40)
	// arguments, which only a ray that has escaped the scene by ~1e18 units can produce
	// (DESIGN.md 1.3 explains why their values are never used).
	unsigned long long sqrt_out_of_domain, divc_out_of_domain; // near field
	unsigned long long far_field;
};
inline Census &census() { static thread_local Census c = {0, 0, 0, 0, 0}; return c; }
inline void census_check_sqrt(float a)
{
	uint32_t u = dm::f2u(a);
	if (a != a || a > 3.402823466e+38f) census().far_field++;
	else if (!(u == 0u || a >= 0x1p-96f)) census().sqrt_out_of_domain++;
}
inline void census_check_divc(float a)
{
	float m = fabsf(a);
	if (a != a || m > 0x1p110f) census().far_field++;
	else if (!(m == 0.f || m >= 0x1p-100f)) census().divc_out_of_domain++;
}
inline void census_check_plane(float a, float c)
{
	float m = fabsf(a);
	if (a != a || m > 0x1p40f) census().far_field++;
	else if (!(m == 0.f || m >= 0x1p-60f) || !(c >= 1e-20f && c <= 2.f))
	{
		census().divc_out_of_domain++;
		if (getenv("ORC_CENSUS_VERBOSE")) fprintf(stderr, "plane out of domain: height %a (%g) denom %a\n", a, a, c);
	}
}
// division by a ray-direction component through a per-ray reciprocal (cube_sea's cell guard in the
// kernels): exact for numerators 0 or 2^-60..2^40 whenever the kernels take that route (|den| >= 2^-60)
inline void census_check_raydiv(float a, float c)
{
	if (!(fabsf(c) >= 0x1p-60f)) return; // the kernels keep the IEEE division here
	float m = fabsf(a);
	if (a != a || m > 0x1p40f) census().far_field++;
	else if (!(m == 0.f || m >= 0x1p-60f))
	{
		census().divc_out_of_domain++;
		if (getenv("ORC_CENSUS_VERBOSE")) fprintf(stderr, "ray division out of domain: numerator %a (%g) den %a\n", a, a, c);
	}
}
#define ORC_COUNT(n) (census().flops += (n))
#define ORC_COUNT_T() (census().flops += 1, census().transc += 1)
#define ORC_CHECK_SQRT(a) census_check_sqrt(a)
#define ORC_CHECK_DIVC(a) census_check_divc(a)
#define ORC_CHECK_PLANE(a, c) census_check_plane(a, c)
#define ORC_CHECK_RAYDIV(a, c) census_check_raydiv(a, c)
struct real
{
	float v;
	real() : v(0.f) {}
	real(float f) : v(f) {}
	real(double f) : v((float)f) {}
	real(int f) : v((float)f) {}
};
inline float val(real a) { return a.v; }
inline real operator+(real a, real b) { ORC_COUNT(1); return real(a.v + b.v); }
inline real operator-(real a, real b) { ORC_COUNT(1); return real(a.v - b.v); }
inline real operator*(real a, real b) { ORC_COUNT(1); return real(a.v * b.v); }
inline real operator/(real a, real b) { ORC_COUNT(1); return real(a.v / b.v); }
inline real operator-(real a) { return real(-a.v); }
inline bool operator<(real a, real b) { ORC_COUNT(1); return a.v < b.v; }
inline bool operator>(real a, real b) { ORC_COUNT(1); return a.v > b.v; }
inline bool operator<=(real a, real b) { ORC_COUNT(1); return a.v <= b.v; }
inline bool operator>=(real a, real b) { ORC_COUNT(1); return a.v >= b.v; }
inline bool operator==(real a, real b) { ORC_COUNT(1); return a.v == b.v; }
inline bool operator!=(real a, real b) { ORC_COUNT(1); return a.v != b.v; }
inline real &operator+=(real &a, real b) { a = a + b; return a; }
inline real &operator-=(real &a, real b) { a = a - b; return a; }
inline real &operator*=(real &a, real b) { a = a * b; return a; }
inline real &operator/=(real &a, real b) { a = a / b; return a; }
#else
typedef float real;
inline float val(real a) { return a; }
#define ORC_COUNT(n) ((void)0)
#define ORC_COUNT_T() ((void)0)
#define ORC_CHECK_SQRT(a) ((void)0)
#define ORC_CHECK_DIVC(a) ((void)0)
#define ORC_CHECK_PLANE(a, c) ((void)0)
#define ORC_CHECK_RAYDIV(a, c) ((void)0)
#endif

typedef unsigned int uint;

// ---- scalar intrinsics -------------------------------------------------------------
inline real r_fma(real a, real b, real c) { ORC_COUNT(2); return real(fmaf(val(a), val(b), val(c))); }
inline real r_abs(real a) { return real(fabsf(val(a))); }
inline real r_floor(real a) { ORC_COUNT(1); return real(floorf(val(a))); }
inline real r_trunc(real a) { ORC_COUNT(1); return real(truncf(val(a))); }
// HLSL round() is round-half-to-even (a-T.1)
inline real r_round(real a) { ORC_COUNT(1); return real(rintf(val(a))); }
inline real r_sqrt(real a) { ORC_COUNT(1); ORC_CHECK_SQRT(val(a)); return real(sqrtf(val(a))); }
inline real r_rsqrt(real a) { ORC_COUNT(1); ORC_CHECK_SQRT(val(a)); return real(1.0f / sqrtf(val(a))); }
// a / c where c is a scene constant: plain IEEE division here; the census build records
// numerators outside the range on which the kernels' div_c is verified
inline real r_div_const(real a, real c) { ORC_CHECK_DIVC(val(a)); return a / c; }
inline real r_sin(real a) { ORC_COUNT_T(); return real(dm::sinf_det(val(a))); }
inline real r_cos(real a) { ORC_COUNT_T(); return real(dm::cosf_det(val(a))); }
inline real r_atan2(real y, real x) { ORC_COUNT_T(); return real(dm::atan2f_det(val(y), val(x))); }
inline real r_exp2(real a) { ORC_COUNT_T(); return real(dm::exp2f_det(val(a))); }
inline real r_log2(real a) { ORC_COUNT_T(); return real(dm::log2f_det(val(a))); }
inline real r_pow(real x, real y) { ORC_COUNT_T(); ORC_COUNT_T(); ORC_COUNT(1); return real(dm::powf_det(val(x), val(y))); }

// IEEE minNum/maxNum with -0 < +0 (a-T.5)
inline real r_min(real a_, real b_)
{
	ORC_COUNT(1);
	float a = val(a_), b = val(b_);
	if (a != a) return real(b);
	if (b != b) return real(a);
	if (a < b) return real(a);
	if (b < a) return real(b);
	return real(std::signbit(a) ? a : b);
}
inline real r_max(real a_, real b_)
{
	ORC_COUNT(1);
	float a = val(a_), b = val(b_);
	if (a != a) return real(b);
	if (b != b) return real(a);
	if (a > b) return real(a);
	if (b > a) return real(b);
	return real(std::signbit(a) ? b : a);
}
inline real r_saturate(real a) { return r_min(r_max(a, real(0.f)), real(1.f)); }
inline real r_clamp(real a, real lo, real hi) { return r_min(r_max(a, lo), hi); }
// step(edge, x) = x >= edge (a-T.3)
inline real r_step(real edge, real x) { return (x >= edge) ? real(1.f) : real(0.f); }
inline real r_sign(real a) { return (a > real(0.f)) ? real(1.f) : ((a < real(0.f)) ? real(-1.f) : real(0.f)); }
inline real r_frac(real a) { return a - r_floor(a); }
inline real r_lerp(real a, real b, real t) { return r_fma(t, b - a, a); }
// FXC expansion of fmod (a-T.2): r = a/b; f = frac(|r|); (r >= -r ? f : -f) * b
inline real r_fmod(real a, real b)
{
	real r = a / b;
	real f = r_frac(r_abs(r));
	real s = (r >= -r) ? f : -f;
	return s * b;
}
// the same with a scene-constant divisor (see r_div_const)
inline real r_fmod_const(real a, real b)
{
	real r = r_div_const(a, b);
	real f = r_frac(r_abs(r));
	real s = (r >= -r) ? f : -f;
	return s * b;
}
// modf: integer part truncated toward zero, returns the signed fractional part
inline real r_modf(real a, real &ip) { ip = r_trunc(a); return a - ip; }
// float -> int, truncating, saturating like D3D ftoi (a-T.6)
inline int r_ftoi(real a_)
{
	float a = val(a_);
	if (a != a) return 0;
	if (a >= 2147483648.f) return 2147483647;
	if (a <= -2147483648.f) return (int)0x80000000;
	return (int)a;
}

// ---- vectors -----------------------------------------------------------------------
struct float2
{
	real x, y;
	float2() {}
	float2(real x_, real y_) : x(x_), y(y_) {}
	explicit float2(real s) : x(s), y(s) {}
};
struct float3
{
	real x, y, z;
	float3() {}
	float3(real x_, real y_, real z_) : x(x_), y(y_), z(z_) {}
	explicit float3(real s) : x(s), y(s), z(s) {}
	float3(float2 a, real z_) : x(a.x), y(a.y), z(z_) {}
};
struct float4
{
	real x, y, z, w;
	float4() {}
	float4(real x_, real y_, real z_, real w_) : x(x_), y(y_), z(z_), w(w_) {}
	explicit float4(real s) : x(s), y(s), z(s), w(s) {}
	float4(float3 a, real w_) : x(a.x), y(a.y), z(a.z), w(w_) {}
	float4(float2 a, float2 b) : x(a.x), y(a.y), z(b.x), w(b.y) {}
	float4(float2 a, real z_, real w_) : x(a.x), y(a.y), z(z_), w(w_) {}
	float3 xyz() const { return float3(x, y, z); }
};

#define ORC_VEC_OPS2(OP) \
	inline float2 operator OP(float2 a, float2 b) { return float2(a.x OP b.x, a.y OP b.y); } \
	inline float2 operator OP(float2 a, real b) { return float2(a.x OP b, a.y OP b); } \
	inline float2 operator OP(real a, float2 b) { return float2(a OP b.x, a OP b.y); }
#define ORC_VEC_OPS3(OP) \
	inline float3 operator OP(float3 a, float3 b) { return float3(a.x OP b.x, a.y OP b.y, a.z OP b.z); } \
	inline float3 operator OP(float3 a, real b) { return float3(a.x OP b, a.y OP b, a.z OP b); } \
	inline float3 operator OP(real a, float3 b) { return float3(a OP b.x, a OP b.y, a OP b.z); }
#define ORC_VEC_OPS4(OP) \
	inline float4 operator OP(float4 a, float4 b) { return float4(a.x OP b.x, a.y OP b.y, a.z OP b.z, a.w OP b.w); } \
	inline float4 operator OP(float4 a, real b) { return float4(a.x OP b, a.y OP b, a.z OP b, a.w OP b); } \
	inline float4 operator OP(real a, float4 b) { return float4(a OP b.x, a OP b.y, a OP b.z, a OP b.w); }
ORC_VEC_OPS2(+) ORC_VEC_OPS2(-) ORC_VEC_OPS2(*) ORC_VEC_OPS2(/)
ORC_VEC_OPS3(+) ORC_VEC_OPS3(-) ORC_VEC_OPS3(*) ORC_VEC_OPS3(/)
ORC_VEC_OPS4(+) ORC_VEC_OPS4(-) ORC_VEC_OPS4(*) ORC_VEC_OPS4(/)
inline float2 operator-(float2 a) { return float2(-a.x, -a.y); }
inline float3 operator-(float3 a) { return float3(-a.x, -a.y, -a.z); }
inline float4 operator-(float4 a) { return float4(-a.x, -a.y, -a.z, -a.w); }

#define ORC_MAP2(NAME, F) inline float2 NAME(float2 a) { return float2(F(a.x), F(a.y)); }
#define ORC_MAP3(NAME, F) inline float3 NAME(float3 a) { return float3(F(a.x), F(a.y), F(a.z)); }
#define ORC_MAP4(NAME, F) inline float4 NAME(float4 a) { return float4(F(a.x), F(a.y), F(a.z), F(a.w)); }
ORC_MAP2(v_abs, r_abs) ORC_MAP3(v_abs, r_abs) ORC_MAP4(v_abs, r_abs)
ORC_MAP2(v_floor, r_floor) ORC_MAP3(v_floor, r_floor) ORC_MAP4(v_floor, r_floor)
ORC_MAP2(v_round, r_round) ORC_MAP3(v_round, r_round)
ORC_MAP2(v_frac, r_frac) ORC_MAP3(v_frac, r_frac)
ORC_MAP2(v_saturate, r_saturate) ORC_MAP3(v_saturate, r_saturate)
ORC_MAP2(v_sin, r_sin)

inline float2 v_min(float2 a, float2 b) { return float2(r_min(a.x, b.x), r_min(a.y, b.y)); }
inline float3 v_min(float3 a, float3 b) { return float3(r_min(a.x, b.x), r_min(a.y, b.y), r_min(a.z, b.z)); }
inline float4 v_min(float4 a, float4 b) { return float4(r_min(a.x, b.x), r_min(a.y, b.y), r_min(a.z, b.z), r_min(a.w, b.w)); }
inline float2 v_max(float2 a, float2 b) { return float2(r_max(a.x, b.x), r_max(a.y, b.y)); }
inline float3 v_max(float3 a, float3 b) { return float3(r_max(a.x, b.x), r_max(a.y, b.y), r_max(a.z, b.z)); }
inline float4 v_max(float4 a, float4 b) { return float4(r_max(a.x, b.x), r_max(a.y, b.y), r_max(a.z, b.z), r_max(a.w, b.w)); }
inline float2 v_max(float2 a, real b) { return v_max(a, float2(b)); }
inline float3 v_max(float3 a, real b) { return v_max(a, float3(b)); }
inline float4 v_max(float4 a, real b) { return v_max(a, float4(b)); }
inline float3 v_max(real a, float3 b) { return v_max(float3(a), b); }
inline float2 v_step(real e, float2 x) { return float2(r_step(e, x.x), r_step(e, x.y)); }
inline float3 v_step(real e, float3 x) { return float3(r_step(e, x.x), r_step(e, x.y), r_step(e, x.z)); }
inline float3 v_step(float3 e, float3 x) { return float3(r_step(e.x, x.x), r_step(e.y, x.y), r_step(e.z, x.z)); }
inline float4 v_step(float4 e, real x) { return float4(r_step(e.x, x), r_step(e.y, x), r_step(e.z, x), r_step(e.w, x)); }
inline float2 v_clamp(float2 a, float2 lo, float2 hi) { return float2(r_clamp(a.x, lo.x, hi.x), r_clamp(a.y, lo.y, hi.y)); }
inline float3 v_clamp(float3 a, float3 lo, float3 hi) { return float3(r_clamp(a.x, lo.x, hi.x), r_clamp(a.y, lo.y, hi.y), r_clamp(a.z, lo.z, hi.z)); }

inline real dot(float2 a, float2 b) { return r_fma(a.y, b.y, a.x * b.x); }
inline real dot(float3 a, float3 b) { return r_fma(a.z, b.z, r_fma(a.y, b.y, a.x * b.x)); }
inline real dot(float4 a, float4 b) { return r_fma(a.w, b.w, r_fma(a.z, b.z, r_fma(a.y, b.y, a.x * b.x))); }
inline real length(float2 a) { return r_sqrt(dot(a, a)); }
inline real length(float3 a) { return r_sqrt(dot(a, a)); }
// normalize(v) = v * rsqrt(dot(v, v)) (a-T.4)
inline float2 normalize(float2 a) { return a * r_rsqrt(dot(a, a)); }
inline float3 normalize(float3 a) { return a * r_rsqrt(dot(a, a)); }
inline float2 lerp(float2 a, float2 b, real t) { return float2(r_lerp(a.x, b.x, t), r_lerp(a.y, b.y, t)); }
inline float3 lerp(float3 a, float3 b, real t) { return float3(r_lerp(a.x, b.x, t), r_lerp(a.y, b.y, t), r_lerp(a.z, b.z, t)); }
inline float2 lerp(float2 a, float2 b, float2 t) { return float2(r_lerp(a.x, b.x, t.x), r_lerp(a.y, b.y, t.y)); }
// a + b*s per component, fused
inline float3 mad(float3 b, real s, float3 a) { return float3(r_fma(b.x, s, a.x), r_fma(b.y, s, a.y), r_fma(b.z, s, a.z)); }
inline bool any(real a) { return a != real(0.f); }
inline bool any(float3 a) { return a.x != real(0.f) || a.y != real(0.f) || a.z != real(0.f); }

// reflect(i, n) = i - 2*dot(i, n)*n
inline float3 reflect(float3 i, float3 n)
{
	real k = real(2.f) * dot(i, n);
	return float3(r_fma(-k, n.x, i.x), r_fma(-k, n.y, i.y), r_fma(-k, n.z, i.z));
}
// refract(i, n, eta): k = 1 - eta^2 (1 - dot(n,i)^2); k < 0 ? 0 : eta*i - (eta*dot(n,i) + sqrt k)*n
inline float3 refract(float3 i, float3 n, real eta)
{
	real d = dot(n, i);
	real k = r_fma(-(eta * eta), r_fma(-d, d, real(1.f)), real(1.f));
	if (k < real(0.f)) return float3(real(0.f));
	real s = r_fma(eta, d, r_sqrt(k));
	return float3(r_fma(-s, n.x, eta * i.x), r_fma(-s, n.y, eta * i.y), r_fma(-s, n.z, eta * i.z));
}

} // namespace orc
