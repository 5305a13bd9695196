// oracle/postprocess.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates the consumer of the hot path's output: HDR::process
// (Engine/Postprocessing.cpp:130-174) = bright-pass + 33-tap stride-2 horizontal Gaussian
// (Engine/shader/bloom.hlsl:14-27), 33-tap stride-2 vertical Gaussian (:29-38), tone map
// (Engine/shader/pshader_hdr.hlsl:16-26).  All three textures are R16G16B16A16_FLOAT
// (Postprocessing.cpp:23), the back buffer is R8G8B8A8_UNORM (Graphics.cpp:65).
// Out-of-range texel reads return 0 (Texture2D operator[] semantics); the linear sampler of
// the tone-map pass reads texel centres, i.e. the texels themselves.
// Arithmetic: fp16 storage with round-to-nearest-even, fp32 arithmetic in source order, the
// tap sums accumulate i = -16 .. +16 sequentially, exp(x) = exp2(x * log2 e) (D3D lowering).
#pragma once
#include "hlsl.h"

#include <cstdint>
#include <vector>

namespace orc {
namespace post {

inline float half_to_float(uint16_t h)
{
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t exp = (h >> 10) & 0x1fu;
	uint32_t man = h & 0x3ffu;
	if (exp == 0)
	{
		if (man == 0) return dm::u2f(sign);
		// subnormal half: value = man * 2^-24
		float v = (float)man * 5.9604644775390625e-8f;
		return dm::u2f(dm::f2u(v) | sign);
	}
	if (exp == 31) return dm::u2f(sign | 0x7f800000u | (man << 13));
	return dm::u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

// round-to-nearest-even, overflow to infinity, NaN kept (what v_cvt_f16_f32 does)
inline uint16_t float_to_half(float f)
{
	uint32_t u = dm::f2u(f);
	uint32_t sign = (u >> 16) & 0x8000u;
	uint32_t a = u & 0x7fffffffu;
	if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (a > 0x7f800000u ? (0x200u | ((a >> 13) & 0x3ffu)) : 0u));
	if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); // >= 65520 rounds to infinity
	if (a < 0x33000001u) return (uint16_t)sign;             // <= 2^-25 rounds to zero (ties to even)
	uint32_t exp = a >> 23;
	uint32_t man = (a & 0x7fffffu) | 0x800000u;
	if (exp < 113u)
	{
		// subnormal result: shift so that the unit is 2^-24
		uint32_t shift = 126u - exp; // 14 .. 24
		uint32_t q = man >> shift;
		uint32_t rem = man & ((1u << shift) - 1u);
		uint32_t half = 1u << (shift - 1u);
		if (rem > half || (rem == half && (q & 1u))) q++;
		return (uint16_t)(sign | q);
	}
	uint32_t q = ((exp - 112u) << 10) | ((man >> 13) & 0x3ffu);
	uint32_t rem = man & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
	return (uint16_t)(sign | q);
}

// bloom.hlsl:5-12
static const float coeffs[17] = {0.070771f, 0.069674f, 0.066483f, 0.061487f, 0.055116f, 0.047886f, 0.040324f, 0.032912f, 0.026035f,
	0.019962f, 0.014834f, 0.010685f, 0.007459f, 0.005047f, 0.003310f, 0.002104f, 0.001296f};

struct Image16
{
	int w, h;
	const uint16_t *px; // RGBA16F, row-major
	float4 at(int x, int y) const
	{
		if (x < 0 || y < 0 || x >= w || y >= h) return float4(real(0.f));
		const uint16_t *p = px + 4 * ((size_t)y * w + x);
		return float4(real(half_to_float(p[0])), real(half_to_float(p[1])), real(half_to_float(p[2])), real(half_to_float(p[3])));
	}
};

inline void store16(uint16_t *p, float4 v)
{
	p[0] = float_to_half(val(v.x));
	p[1] = float_to_half(val(v.y));
	p[2] = float_to_half(val(v.z));
	p[3] = float_to_half(val(v.w));
}

// bloom.hlsl:14-27 (cs_main1)
inline void bloom_horizontal(const Image16 &scene, uint16_t *out)
{
	for (int y = 0; y < scene.h; ++y)
		for (int x = 0; x < scene.w; ++x)
		{
			float4 sum = float4(real(0.f));
			for (int i = -16; i <= 16; ++i)
			{
				float4 col = scene.at(x + 2 * i, y);
				real brightness = dot(col.xyz(), float3(real(0.2126f), real(0.7152f), real(0.0722f)));
				real factor = r_saturate((r_saturate(brightness) - real(0.75f)) * real(4.f));
				col = col * factor;
				sum = sum + col * real(coeffs[i < 0 ? -i : i]);
			}
			store16(out + 4 * ((size_t)y * scene.w + x), sum * real(2.f));
		}
}

// bloom.hlsl:29-38 (cs_main2)
inline void bloom_vertical(const Image16 &in, uint16_t *out)
{
	for (int y = 0; y < in.h; ++y)
		for (int x = 0; x < in.w; ++x)
		{
			float4 sum = float4(real(0.f));
			for (int i = -16; i <= 16; ++i)
				sum = sum + in.at(x, y + 2 * i) * real(coeffs[i < 0 ? -i : i]);
			store16(out + 4 * ((size_t)y * in.w + x), sum * real(2.f));
		}
}

inline real exp_d3d(real x) { return r_exp2(x * real(1.44269504088896340736f)); }

// R8G8B8A8_UNORM conversion of D3D: NaN -> 0, clamp to [0,1], * 255, + 0.5, truncate
inline uint8_t to_unorm8(real v)
{
	real c = r_saturate(v);
	return (uint8_t)r_ftoi(c * real(255.f) + real(0.5f));
}

// pshader_hdr.hlsl:16-26
inline void tonemap(const Image16 &scene, const Image16 &bloom, uint8_t *out)
{
	for (int y = 0; y < scene.h; ++y)
		for (int x = 0; x < scene.w; ++x)
		{
			float4 scene_color = scene.at(x, y);
			float4 total = scene_color + bloom.at(x, y);
			const real exposure = 1.f;
			float4 e = -total * exposure;
			float4 ldr = real(1.f) - float4(exp_d3d(e.x), exp_d3d(e.y), exp_d3d(e.z), exp_d3d(e.w));
			real a = scene_color.w;
			float4 o = float4(r_lerp(scene_color.x, ldr.x, a), r_lerp(scene_color.y, ldr.y, a), r_lerp(scene_color.z, ldr.z, a), r_lerp(scene_color.w, ldr.w, a));
			uint8_t *p = out + 4 * ((size_t)y * scene.w + x);
			p[0] = to_unorm8(o.x);
			p[1] = to_unorm8(o.y);
			p[2] = to_unorm8(o.z);
			p[3] = to_unorm8(o.w);
		}
}

} // namespace post
} // namespace orc
