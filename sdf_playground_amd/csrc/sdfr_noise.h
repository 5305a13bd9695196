// sdfr_noise.h -- PCG hash, simplex noise in 2, 3 and 4 dimensions and 4-octave turbulence for the shade kernels.
//
// Same functions as the reference's Engine/shader/noise.hlsl (hash :6-16, grad4 :124-140, snoise(float2)
// :142-203, snoise(float3) :205-300, snoise(float4) :304-433, turbulence :473-476), written corner-by-corner
// in scalars so that no float4 temporaries have to stay live: the simplex corners are independent until the
// final weighted sum.  Operation order per value is the one of the HLSL source.
#pragma once
#include "sdfr_math.h"

namespace sdfr {

SDF_HD uint32_t pcg_hash(uint32_t v)
{
	uint32_t state = v * 747796405u + 2891336453u;
	uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
	return (word >> 22u) ^ word;
}
SDF_HD float pcg_hashf(uint32_t v) { return (float)pcg_hash(v) / (float)0xFFFFFFFFu; }

SDF_HD float noise_mod289(float x) { return x - floor1(x * 0.00346020761245674740484429065744f) * 289.0f; }
SDF_HD float noise_permute(float x) { return noise_mod289(x * x * 34.0f + x); }

// gradient of one simplex corner dotted with the corner offset `xc`, times the corner's
// falloff weight: returns (m^4, dot(grad, xc)) as (x, y)
SDF_HD vec2 simplex_corner(float p, vec3 xc)
{
	const float n_ = 0.142857142857f;
	const float ns_x = n_ * 2.0f - 0.0f, ns_y = n_ * 0.5f - 1.0f, ns_z = n_ * 1.0f - 0.0f;
	float j = p - 49.0f * floor1(p * ns_z * ns_z);
	float xq = floor1(j * ns_z);
	float yq = floor1(j - 7.0f * xq);
	float gx = xq * ns_x + ns_y;
	float gy = yq * ns_x + ns_y;
	float h = 1.0f - abs1(gx) - abs1(gy);
	float sx = floor1(gx) * 2.0f + 1.0f;
	float sy = floor1(gy) * 2.0f + 1.0f;
	float sh = -step1(h, 0.0f);
	vec3 g = V3(gx + sx * sh, gy + sy * sh, h);
	g = g * rsqrt1(dot(g, g));
	float m = max1(0.6f - dot(xc, xc), 0.0f);
	m = m * m;
	return V2(m * m, dot(g, xc));
}

SDF_HD float snoise3(vec3 v)
{
	const float Cx = 0.166666666666666667f, Cy = 0.333333333333333333f;
	// skew to the simplex grid
	vec3 i = floor(v + dot(v, V3s(Cy)));
	vec3 x0 = v - i + dot(i, V3s(Cx));
	// rank the components to pick the traversal order of the simplex
	vec3 g = V3(step1(x0.y, x0.x), step1(x0.z, x0.y), step1(x0.x, x0.z));
	vec3 l = 1.f - g;
	vec3 lz = V3(l.z, l.x, l.y);
	vec3 i1 = min(g, lz);
	vec3 i2 = max(g, lz);
	vec3 x1 = x0 - i1 + Cx;
	vec3 x2 = x0 - i2 + Cy;
	vec3 x3 = x0 - 0.5f;

	i = V3(noise_mod289(i.x), noise_mod289(i.y), noise_mod289(i.z));
	// hashed gradient index of each corner; the "+ 0" of corner 0 is the identity because
	// mod289 never returns -0
	float p0 = noise_permute(noise_permute(noise_permute(i.z) + i.y) + i.x);
	float p1 = noise_permute(noise_permute(noise_permute(i.z + i1.z) + i.y + i1.y) + i.x + i1.x);
	float p2 = noise_permute(noise_permute(noise_permute(i.z + i2.z) + i.y + i2.y) + i.x + i2.x);
	float p3 = noise_permute(noise_permute(noise_permute(i.z + 1.0f) + i.y + 1.0f) + i.x + 1.0f);

	vec2 c0 = simplex_corner(p0, x0);
	vec2 c1 = simplex_corner(p1, x1);
	vec2 c2 = simplex_corner(p2, x2);
	vec2 c3 = simplex_corner(p3, x3);
	return 42.0f * dot(V4(c0.x, c1.x, c2.x, c3.x), V4(c0.y, c1.y, c2.y, c3.y));
}

// ---- 2-D (noise.hlsl:142-203) ------------------------------------------------------------------------------
// one corner: the hashed index p picks one of 41 gradients on a diamond; returns (m^4 times the approximate
// normalisation, gradient . offset)
SDF_HD vec2 simplex2_corner(float p, float ox, float oy)
{
	float m = max1(0.5f - fma1(oy, oy, ox * ox), 0.0f);
	m = m * m;
	m = m * m;
	const float x = 2.0f * frac1(p * 0.024390243902439f) - 1.0f;
	const float h = abs1(x) - 0.5f;
	const float a0 = x - floor1(x + 0.5f);
	m = m * (1.79284291400159f - 0.85373472095314f * (a0 * a0 + h * h));
	return V2(m, a0 * ox + h * oy);
}

SDF_HD float snoise2(vec2 v)
{
	const float Cx = 0.211324865405187f, Cy = 0.366025403784439f, Cz = -0.577350269189626f;
	const float s = dot(v, V2(Cy, Cy));
	float ix = floor1(v.x + s), iy = floor1(v.y + s);
	const float t = dot(V2(ix, iy), V2(Cx, Cx));
	const float x0 = v.x - ix + t, y0 = v.y - iy + t;
	// the middle corner lies one step along the larger component
	const float sx = x0 > y0 ? 1.0f : 0.0f, sy = 1.0f - sx;
	const float x1 = x0 + Cx - sx, y1 = y0 + Cx - sy;
	const float x2 = x0 + Cz, y2 = y0 + Cz;
	ix = noise_mod289(ix);
	iy = noise_mod289(iy);
	// `iy + 0` of the first corner is the identity: mod289 never returns -0
	const float p0 = noise_permute(noise_permute(iy) + ix);
	const float p1 = noise_permute(noise_permute(iy + sy) + ix + sx);
	const float p2 = noise_permute(noise_permute(iy + 1.0f) + ix + 1.0f);
	const vec2 c0 = simplex2_corner(p0, x0, y0);
	const vec2 c1 = simplex2_corner(p1, x1, y1);
	const vec2 c2 = simplex2_corner(p2, x2, y2);
	return 130.0f * dot(V3(c0.x, c1.x, c2.x), V3(c0.y, c1.y, c2.y));
}

// ---- 4-D (noise.hlsl:124-140, 304-433) -----------------------------------------------------------------------
// gradient j of the 7 x 7 x 6 points on a cube, folded onto the 4-cross polytope (not normalised)
SDF_HD vec4 noise_grad4(float j, float ipx, float ipy, float ipz)
{
	float gx = floor1(frac1(j * ipx) * 7.0f) * ipz - 1.0f;
	float gy = floor1(frac1(j * ipy) * 7.0f) * ipz - 1.0f;
	float gz = floor1(frac1(j * ipz) * 7.0f) * ipz - 1.0f;
	const float gw = 1.5f - dot(V3(abs1(gx), abs1(gy), abs1(gz)), V3s(1.0f));
	const float below = gw < 0.f ? 1.0f : 0.0f;
	gx = gx - sign1(gx) * below;
	gy = gy - sign1(gy) * below;
	gz = gz - sign1(gz) * below;
	return V4(gx, gy, gz, gw);
}
// one corner: (falloff^4, normalised gradient . offset)
SDF_HD vec2 simplex4_corner(float j, vec4 xc)
{
	vec4 g = noise_grad4(j, 0.003401360544217687075f, 0.020408163265306122449f, 0.142857142857142857143f); // 1/294, 1/49, 1/7
	g = g * rsqrt1(dot(g, g));
	float m = max1(0.6f - dot(xc, xc), 0.0f);
	m = m * m;
	return V2(m * m, dot(g, xc));
}

SDF_HD float snoise4(vec4 v)
{
	const float G4 = 0.138196601125011f, G4x2 = 0.276393202250021f, G4x3 = 0.414589803375032f, G4x4m1 = -0.447213595499958f;
	const float s = dot(v, V4(0.309016994374947451f, 0.309016994374947451f, 0.309016994374947451f, 0.309016994374947451f));
	vec4 i = floor(v + s);
	const vec4 x0 = v - i + dot(i, V4(G4, G4, G4, G4));
	// rank of each component among the four (3 = largest; ties as step(): x >= edge counts for the earlier one)
	const float xy = step1(x0.y, x0.x), xz = step1(x0.z, x0.x), xw = step1(x0.w, x0.x);
	const float yz = step1(x0.z, x0.y), yw = step1(x0.w, x0.y), zw = step1(x0.w, x0.z);
	const vec4 rank = V4(xy + xz + xw, (1.0f - xy) + (yz + yw), ((1.0f - xz) + (1.0f - yz)) + zw, ((1.0f - xw) + (1.0f - yw)) + (1.0f - zw));
	// corner k steps along the components of rank >= 4 - k
	const vec4 i3 = V4(sat1(rank.x), sat1(rank.y), sat1(rank.z), sat1(rank.w));
	const vec4 i2 = V4(sat1(rank.x - 1.0f), sat1(rank.y - 1.0f), sat1(rank.z - 1.0f), sat1(rank.w - 1.0f));
	const vec4 i1 = V4(sat1(rank.x - 2.0f), sat1(rank.y - 2.0f), sat1(rank.z - 2.0f), sat1(rank.w - 2.0f));
	const vec4 x1 = x0 - i1 + G4;
	const vec4 x2 = x0 - i2 + G4x2;
	const vec4 x3 = x0 - i3 + G4x3;
	const vec4 x4 = x0 + G4x4m1;
	i = V4(noise_mod289(i.x), noise_mod289(i.y), noise_mod289(i.z), noise_mod289(i.w));
	const float j0 = noise_permute(noise_permute(noise_permute(noise_permute(i.w) + i.z) + i.y) + i.x);
	const float j1 = noise_permute(noise_permute(noise_permute(noise_permute(i.w + i1.w) + i.z + i1.z) + i.y + i1.y) + i.x + i1.x);
	const float j2 = noise_permute(noise_permute(noise_permute(noise_permute(i.w + i2.w) + i.z + i2.z) + i.y + i2.y) + i.x + i2.x);
	const float j3 = noise_permute(noise_permute(noise_permute(noise_permute(i.w + i3.w) + i.z + i3.z) + i.y + i3.y) + i.x + i3.x);
	const float j4 = noise_permute(noise_permute(noise_permute(noise_permute(i.w + 1.0f) + i.z + 1.0f) + i.y + 1.0f) + i.x + 1.0f);
	const vec2 c0 = simplex4_corner(j0, x0);
	const vec2 c1 = simplex4_corner(j1, x1);
	const vec2 c2 = simplex4_corner(j2, x2);
	const vec2 c3 = simplex4_corner(j3, x3);
	const vec2 c4 = simplex4_corner(j4, x4);
	return 49.0f * (dot(V3(c0.x, c1.x, c2.x), V3(c0.y, c1.y, c2.y)) + dot(V2(c3.x, c4.x), V2(c3.y, c4.y)));
}

SDF_HD float turbulence3(vec3 p)
{
	return div_c((snoise3(p) + snoise3(p * 2.f) / 2.f + snoise3(p * 4.f) / 4.f + snoise3(p * 8.f) / 8.f) * 8.f, 15.f, 1.0f / 15.f);
}

} // namespace sdfr
