// oracle/noise.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates Engine/shader/noise.hlsl: PCG hash (:6-16), Ashima simplex noise in 2, 3 and 4
// dimensions (:78-120 helpers, :124-140 grad4, :142-203 snoise(float2), :205-300 snoise(float3),
// :304-433 snoise(float4)), turbulence (:473-476).
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
inline float2 mod289(float2 x) { return float2(mod289(x.x), mod289(x.y)); }
inline float3 mod289(float3 x) { return float3(mod289(x.x), mod289(x.y), mod289(x.z)); }
inline float4 mod289(float4 x) { return float4(mod289(x.x), mod289(x.y), mod289(x.z), mod289(x.w)); }
// noise.hlsl:101-120: mod289(x*x*34 + x)
inline real permute(real x) { return mod289(x * x * real(34.0f) + x); }
inline float3 permute(float3 x) { return float3(permute(x.x), permute(x.y), permute(x.z)); }
inline float4 permute(float4 x) { return float4(permute(x.x), permute(x.y), permute(x.z), permute(x.w)); }

// noise.hlsl:124-138 -- a hashed gradient on the 4-D cross polytope; `(p.w < 0)` is HLSL's bool -> 0 / 1
inline float4 grad4(real j, float4 ip)
{
	const float4 ones = float4(1.0f, 1.0f, 1.0f, -1.0f);
	float4 p;
	float3 q = v_floor(v_frac(j * ip.xyz()) * real(7.0f)) * ip.z - real(1.0f);
	p.x = q.x; p.y = q.y; p.z = q.z;
	p.w = real(1.5f) - dot(v_abs(p.xyz()), ones.xyz());
	const real neg = (p.w < real(0.f)) ? real(1.f) : real(0.f);
	p.x = p.x - r_sign(p.x) * neg;
	p.y = p.y - r_sign(p.y) * neg;
	p.z = p.z - r_sign(p.z) * neg;
	return p;
}

// noise.hlsl:144-201
inline real snoise(float2 v)
{
	const float4 C = float4(
		0.211324865405187f, // (3.0-sqrt(3.0))/6.0
		0.366025403784439f, // 0.5*(sqrt(3.0)-1.0)
		-0.577350269189626f, // -1.0 + 2.0 * C.x
		0.024390243902439f); // 1.0 / 41.0

	// first corner (:154-155)
	float2 i = v_floor(v + dot(v, float2(C.y, C.y)));
	float2 x0 = v - i + dot(i, float2(C.x, C.x));

	// other corners (:163-165); i1 is an int2 in the source: 0 / 1, exact either way
	float2 i1 = (x0.x > x0.y) ? float2(1.0f, 0.0f) : float2(0.0f, 1.0f);
	float4 x12 = float4(x0.x, x0.y, x0.x, x0.y) + float4(C.x, C.x, C.z, C.z);
	x12.x = x12.x - i1.x;
	x12.y = x12.y - i1.y;

	// permutations (:168-173)
	i = mod289(i);
	float3 p = permute(
		permute(i.y + float3(real(0.0f), i1.y, real(1.0f)))
		+ i.x + float3(real(0.0f), i1.x, real(1.0f)));

	float3 m = v_max(real(0.5f) - float3(dot(x0, x0), dot(float2(x12.x, x12.y), float2(x12.x, x12.y)), dot(float2(x12.z, x12.w), float2(x12.z, x12.w))), real(0.0f));
	m = m * m;
	m = m * m;

	// gradients: 41 points over a line, mapped onto a diamond (:189-192)
	float3 x = real(2.0f) * v_frac(p * float3(C.w)) - real(1.0f);
	float3 h = v_abs(x) - real(0.5f);
	float3 ox = v_floor(x + real(0.5f));
	float3 a0 = x - ox;

	// approximate normalisation (:196)
	m = m * (real(1.79284291400159f) - real(0.85373472095314f) * (a0 * a0 + h * h));

	// noise value (:199-202)
	float3 g;
	g.x = a0.x * x0.x + h.x * x0.y;
	g.y = a0.y * x12.x + h.y * x12.y;
	g.z = a0.z * x12.z + h.z * x12.w;
	return real(130.0f) * dot(m, g);
}

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

// noise.hlsl:304-433
inline real snoise(float4 v)
{
	const float4 C = float4(
		0.138196601125011f, // (5 - sqrt(5))/20 G4
		0.276393202250021f, // 2 * G4
		0.414589803375032f, // 3 * G4
		-0.447213595499958f); // -1 + 4 * G4

	// first corner (:314-322)
	float4 i = v_floor(v + dot(v, float4(real(0.309016994374947451f))));
	float4 x0 = v - i + dot(i, float4(C.x));

	// other corners: rank sorting (:327-335); step(edge, x) = x >= edge
	float4 i0;
	float3 isX = v_step(float3(x0.y, x0.z, x0.w), float3(x0.x, x0.x, x0.x));
	float3 isYZ = v_step(float3(x0.z, x0.w, x0.w), float3(x0.y, x0.y, x0.z));
	i0.x = isX.x + isX.y + isX.z;
	i0.y = real(1.0f) - isX.x;
	i0.z = real(1.0f) - isX.y;
	i0.w = real(1.0f) - isX.z;
	i0.y = i0.y + (isYZ.x + isYZ.y);
	i0.z = i0.z + (real(1.0f) - isYZ.x);
	i0.w = i0.w + (real(1.0f) - isYZ.y);
	i0.z = i0.z + isYZ.z;
	i0.w = i0.w + (real(1.0f) - isYZ.z);

	// i0 now holds 0, 1, 2, 3 once each (:338-340)
	float4 i3 = float4(r_saturate(i0.x), r_saturate(i0.y), r_saturate(i0.z), r_saturate(i0.w));
	float4 i2 = float4(r_saturate(i0.x - real(1.0f)), r_saturate(i0.y - real(1.0f)), r_saturate(i0.z - real(1.0f)), r_saturate(i0.w - real(1.0f)));
	float4 i1 = float4(r_saturate(i0.x - real(2.0f)), r_saturate(i0.y - real(2.0f)), r_saturate(i0.z - real(2.0f)), r_saturate(i0.w - real(2.0f)));

	// (:347-350)
	float4 x1 = x0 - i1 + float4(C.x);
	float4 x2 = x0 - i2 + float4(C.y);
	float4 x3 = x0 - i3 + float4(C.z);
	float4 x4 = x0 + float4(C.w);

	// permutations (:353-369)
	i = mod289(i);
	real j0 = permute(permute(permute(permute(i.w) + i.z) + i.y) + i.x);
	float4 j1 = permute(
		permute(
			permute(
				permute(i.w + float4(i1.w, i2.w, i3.w, real(1.0f)))
				+ i.z + float4(i1.z, i2.z, i3.z, real(1.0f)))
			+ i.y + float4(i1.y, i2.y, i3.y, real(1.0f)))
		+ i.x + float4(i1.x, i2.x, i3.x, real(1.0f)));

	// gradients: 7 x 7 x 6 points over a cube, mapped onto a 4-cross polytope (:373-384)
	const float4 ip = float4(0.003401360544217687075f, 0.020408163265306122449f, 0.142857142857142857143f, 0.0f);
	float4 p0 = grad4(j0, ip);
	float4 p1 = grad4(j1.x, ip);
	float4 p2 = grad4(j1.y, ip);
	float4 p3 = grad4(j1.z, ip);
	float4 p4 = grad4(j1.w, ip);

	// normalise gradients (:387-397)
	float4 norm = float4(r_rsqrt(dot(p0, p0)), r_rsqrt(dot(p1, p1)), r_rsqrt(dot(p2, p2)), r_rsqrt(dot(p3, p3)));
	p0 = p0 * norm.x;
	p1 = p1 * norm.y;
	p2 = p2 * norm.z;
	p3 = p3 * norm.w;
	p4 = p4 * r_rsqrt(dot(p4, p4));

	// mix the five corners (:400-432)
	float3 m0 = v_max(real(0.6f) - float3(dot(x0, x0), dot(x1, x1), dot(x2, x2)), real(0.0f));
	float2 m1 = v_max(real(0.6f) - float2(dot(x3, x3), dot(x4, x4)), real(0.0f));
	m0 = m0 * m0;
	m1 = m1 * m1;
	return real(49.0f) * (dot(m0 * m0, float3(dot(p0, x0), dot(p1, x1), dot(p2, x2))) + dot(m1 * m1, float2(dot(p3, x3), dot(p4, x4))));
}

// noise.hlsl:473-476
inline real turbulence(float3 pos)
{
	return r_div_const((snoise(pos) + snoise(pos * real(2.f)) / real(2.f) + snoise(pos * real(4.f)) / real(4.f) + snoise(pos * real(8.f)) / real(8.f)) * real(8.f), real(15.f));
}

} // namespace orc
