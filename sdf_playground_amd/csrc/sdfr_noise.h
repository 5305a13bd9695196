// sdfr_noise.h -- PCG hash, 3-D simplex noise and 4-octave turbulence for the shade kernels.
//
// Same functions as the reference's Engine/shader/noise.hlsl (hash :6-16, snoise(float3)
// :205-300, turbulence :473-476), written corner-by-corner in scalars so that no float4
// temporaries have to stay live: the four simplex corners are independent until the final
// weighted sum.  Operation order per value is the one of the HLSL source.
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

SDF_HD float turbulence3(vec3 p)
{
	return div_c((snoise3(p) + snoise3(p * 2.f) / 2.f + snoise3(p * 4.f) / 4.f + snoise3(p * 8.f) / 8.f) * 8.f, 15.f, 1.0f / 15.f);
}

} // namespace sdfr
