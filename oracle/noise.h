// oracle/noise.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates Engine/shader/noise.hlsl: PCG hash (:6-16), Ashima 3-D simplex noise
// (:78-120 helpers, :205-300 snoise(float3)), turbulence (:473-476).
// Operation order follows the HLSL source expression by expression; dot() is the
// fused helper of hlsl.h, everything else is unfused.
#pragma once
#include "hlsl.h"

namespace orc {

// noise.hlsl:6-11 -- pure uint32 wrap-around arithmetic (a-T.6)
inline uint32_t hash(uint32_t input)
{
	uint32_t state = input * 747796405u + 2891336453u;
	uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
	return (word >> 22u) ^ word;
}

// noise.hlsl:13-16 -- float(hash)/float(0xFFFFFFFF); uint->float is round-to-nearest-even
inline real hashf(uint32_t input)
{
	return real((float)hash(input)) / real((float)0xFFFFFFFFu);
}

// noise.hlsl:76-96
inline real mod289(real x) { return x - r_floor(x * real(0.00346020761245674740484429065744f)) * real(289.0f); }
inline float3 mod289(float3 x) { return float3(mod289(x.x), mod289(x.y), mod289(x.z)); }
inline float4 mod289(float4 x) { return float4(mod289(x.x), mod289(x.y), mod289(x.z), mod289(x.w)); }
// noise.hlsl:101-120: mod289(x*x*34 + x)
inline real permute(real x) { return mod289(x * x * real(34.0f) + x); }
inline float4 permute(float4 x) { return float4(permute(x.x), permute(x.y), permute(x.z), permute(x.w)); }

// noise.hlsl:205-300
inline real snoise(float3 v)
{
	const real Cx = 0.166666666666666667f, Cy = 0.333333333333333333f;
	const real Dx = 0.0f, Dy = 0.5f, Dz = 1.0f, Dw = 2.0f;

	// first corner (:214-215)
	float3 i = v_floor(v + dot(v, float3(Cy)));
	float3 x0 = v - i + dot(i, float3(Cx));

	// other corners (:218-225)
	float3 g = v_step(float3(x0.y, x0.z, x0.x), x0);
	float3 l = real(1.f) - g;
	float3 lzxy = float3(l.z, l.x, l.y);
	float3 i1 = v_min(g, lzxy);
	float3 i2 = v_max(g, lzxy);

	float3 x1 = x0 - i1 + Cx;
	float3 x2 = x0 - i2 + Cy;
	float3 x3 = x0 - Dy;

	// permutations (:228-235)
	i = mod289(i);
	float4 p = permute(
		permute(
			permute(i.z + float4(real(0.f), i1.z, i2.z, real(1.f)))
			+ i.y + float4(real(0.f), i1.y, i2.y, real(1.f)))
		+ i.x + float4(real(0.f), i1.x, i2.x, real(1.f)));

	// gradients (:239-261)
	const real n_ = 0.142857142857f;
	const real ns_x = n_ * Dw - Dx, ns_y = n_ * Dy - Dz, ns_z = n_ * Dz - Dx;

	float4 j = p - real(49.0f) * v_floor(p * ns_z * ns_z);

	float4 x_ = v_floor(j * ns_z);
	float4 y_ = v_floor(j - real(7.0f) * x_);

	float4 x = x_ * ns_x + ns_y;
	float4 y = y_ * ns_x + ns_y;
	float4 h = real(1.0f) - v_abs(x) - v_abs(y);

	float4 b0 = float4(x.x, x.y, y.x, y.y);
	float4 b1 = float4(x.z, x.w, y.z, y.w);

	float4 s0 = v_floor(b0) * real(2.0f) + real(1.0f);
	float4 s1 = v_floor(b1) * real(2.0f) + real(1.0f);
	float4 sh = -v_step(h, real(0.0f));

	float4 a0 = float4(b0.x, b0.z, b0.y, b0.w) + float4(s0.x, s0.z, s0.y, s0.w) * float4(sh.x, sh.x, sh.y, sh.y);
	float4 a1 = float4(b1.x, b1.z, b1.y, b1.w) + float4(s1.x, s1.z, s1.y, s1.w) * float4(sh.z, sh.z, sh.w, sh.w);

	float3 p0 = float3(a0.x, a0.y, h.x);
	float3 p1 = float3(a0.z, a0.w, h.y);
	float3 p2 = float3(a1.x, a1.y, h.z);
	float3 p3 = float3(a1.z, a1.w, h.w);

	// normalise gradients (:269-278)
	p0 = p0 * r_rsqrt(dot(p0, p0));
	p1 = p1 * r_rsqrt(dot(p1, p1));
	p2 = p2 * r_rsqrt(dot(p2, p2));
	p3 = p3 * r_rsqrt(dot(p3, p3));

	// mix (:281-299)
	float4 m = v_max(real(0.6f) - float4(dot(x0, x0), dot(x1, x1), dot(x2, x2), dot(x3, x3)), real(0.0f));
	m = m * m;
	return real(42.0f) * dot(m * m, float4(dot(p0, x0), dot(p1, x1), dot(p2, x2), dot(p3, x3)));
}

// noise.hlsl:473-476
inline real turbulence(float3 pos)
{
	return r_div_const((snoise(pos) + snoise(pos * real(2.f)) / real(2.f) + snoise(pos * real(4.f)) / real(4.f) + snoise(pos * real(8.f)) / real(8.f)) * real(8.f), real(15.f));
}

} // namespace orc
