// sdfr_post.hip -- the consumer of the raymarch output on gfx950: bloom + tone map.
//
// Mirrors the reference's HDR::process (Engine/Postprocessing.cpp:130-174): bright-pass +
// 33-tap stride-2 horizontal Gaussian (Engine/shader/bloom.hlsl:14-27), 33-tap stride-2
// vertical Gaussian (:29-38), exponential tone map blended by the alpha flag
// (Engine/shader/pshader_hdr.hlsl:16-26) into an R8G8B8A8_UNORM image.  The reference runs
// three passes over three RGBA16F textures; here two HBM-bound kernels:
//   k_bloom_h        scene(f16) -> bloom1(f16).  The bright-pass is applied once per input
//                    texel while staging a row segment in LDS (the reference recomputes it
//                    for each of the 33 taps).
//   k_bloom_v_tone   bloom1(f16) + scene(f16) -> ldr(unorm8).  The vertical blur is staged in
//                    LDS tiles and its result is rounded to f16 in registers -- exactly what
//                    storing the reference's bloom2 texture does -- so bloom2 never exists.
// Algorithmic HBM bytes per pixel: 8 + 8 (pass 1), 8 + 8 + 4 (pass 2) = 36 B.
//
// Where the taps are worth computing.  With the reference's arithmetic (33 taps x 4 channels, a multiply
// and an add each: nothing may fuse) a blur costs 264 vector instructions per pixel and the kernels are
// VALU-bound, not HBM-bound.  But the bright-pass lets through only what is brighter than 0.75, so in most
// of a frame every staged texel is zero, and a sum of zeros is +0 whatever the weights (0 * w = +-0, and
// +0 + -0 = +0): a block whose staged tile holds nothing but zeros writes zeros (pass 1) or goes straight
// to the tone map (pass 2) -- the same bytes, at the speed of the memory system.  Pass 1 also leaves one byte
// per 128 pixels of a row saying whether it stored any light there, so a dark tile of pass 2 does not even
// read its bloom texels (which it would read three times over: 96 staged rows for 32 produced).  Blocks that
// do see light run the taps as before.
#include "sdfr_kernels.h"
#include "sdfr_math.h"

#include <hip/hip_fp16.h>

namespace sdfr {

// bloom.hlsl:3-12
constexpr float bloom_coeff(int i)
{
	constexpr float c[17] = {0.070771f, 0.069674f, 0.066483f, 0.061487f, 0.055116f, 0.047886f, 0.040324f, 0.032912f, 0.026035f,
		0.019962f, 0.014834f, 0.010685f, 0.007459f, 0.005047f, 0.003310f, 0.002104f, 0.001296f};
	return c[i < 0 ? -i : i];
}
// The 33 taps are unrolled by template recursion: the weights are constant expressions and
// become literals of the multiplies (an SGPR or constant-memory operand halves the VALU issue
// rate).  STRIDE = distance in float4 elements between two taps of the staged tile.
template <int I, int STRIDE>
struct Taps
{
	static __device__ __forceinline__ void run(vec4 &sum, const float4 *centre)
	{
		constexpr float w = bloom_coeff(I);
		const float4 t = centre[I * STRIDE];
		sum = sum + V4(t.x, t.y, t.z, t.w) * w;
		if constexpr (I < 16) Taps<I + 1, STRIDE>::run(sum, centre);
	}
};

__device__ __forceinline__ vec4 load_half4(const uint2 *img, size_t idx)
{
	const uint2 v = img[idx];
	const __half2 lo = *reinterpret_cast<const __half2 *>(&v.x);
	const __half2 hi = *reinterpret_cast<const __half2 *>(&v.y);
	const float2 a = __half22float2(lo), b = __half22float2(hi);
	return V4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ uint2 pack_half4(vec4 c)
{
	__half2 lo = __floats2half2_rn(c.x, c.y);
	__half2 hi = __floats2half2_rn(c.z, c.w);
	uint2 v;
	v.x = *reinterpret_cast<uint32_t *>(&lo);
	v.y = *reinterpret_cast<uint32_t *>(&hi);
	return v;
}
// value after a round trip through an f16 texture
__device__ __forceinline__ vec4 through_half4(vec4 c)
{
	__half2 lo = __floats2half2_rn(c.x, c.y);
	__half2 hi = __floats2half2_rn(c.z, c.w);
	const float2 a = __half22float2(lo), b = __half22float2(hi);
	return V4(a.x, a.y, b.x, b.y);
}

#define POST_HALO 32 // 16 taps of stride 2 on each side
#define POST_SEG 512 // pixels of a row one block of the horizontal pass produces; its light flag covers them

__device__ __forceinline__ bool is_lit(vec4 c) { return !(c.x == 0.f && c.y == 0.f && c.z == 0.f && c.w == 0.f); } // NaN counts as lit
__device__ __forceinline__ vec4 bright_pass(vec4 c)
{
	const float brightness = dot(V3(c.x, c.y, c.z), V3(0.2126f, 0.7152f, 0.0722f));
	const float factor = sat1((sat1(brightness) - 0.75f) * 4.f);
	return c * factor;
}
__device__ __forceinline__ vec4 unpack_half4(uint32_t lo, uint32_t hi)
{
	const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&lo)), b = __half22float2(*reinterpret_cast<const __half2 *>(&hi));
	return V4(a.x, a.y, b.x, b.y);
}

// ---- pass 1: bright-pass + horizontal blur ---------------------------------------------------------
// One block per POST_SEG pixels of a row, two neighbouring pixels per thread (16-byte loads and stores where
// the row is 16-byte aligned).  The taps are two texels apart, so an output pixel only ever reads texels of
// its own parity: even and odd texels are staged in separate arrays and a wave's 64 tap reads are 64
// consecutive float4 (no bank conflict).  A wave runs its taps only if one of the 64-texel chunks its
// windows reach into holds light.  flags[y * n + x / POST_FLAG_PIXELS] (one byte per wave: 128 pixels of a row)
// says whether anything but zeros was stored there: the vertical pass reads it instead of the bloom texels
// wherever it is clear.
#define POST_CHUNKS ((POST_SEG + 2 * POST_HALO) / 64)
#define POST_FLAG_PIXELS 128
__global__ __launch_bounds__(POST_SEG / 2) void k_bloom_h(const uint2 *__restrict__ scene, uint2 *__restrict__ bloom1, unsigned char *__restrict__ flags,
	int width, int height)
{
	__shared__ float4 even[(POST_SEG + 2 * POST_HALO) / 2], odd[(POST_SEG + 2 * POST_HALO) / 2];
	__shared__ int chunk_lit[POST_CHUNKS];
	if (threadIdx.x < POST_CHUNKS) chunk_lit[threadIdx.x] = 0;
	__syncthreads();
	const int y = blockIdx.y;
	const int x0 = blockIdx.x * POST_SEG;
	const size_t row = (size_t)y * width;
	const bool aligned = ((row + (size_t)x0) & 1) == 0; // POST_HALO is even: pairs starting at x0 - POST_HALO + 2 k are 16-byte aligned
	for (int i = 2 * (int)threadIdx.x; i < POST_SEG + 2 * POST_HALO; i += POST_SEG)
	{
		const int x = x0 - POST_HALO + i;
		vec4 c0 = V4(0.f, 0.f, 0.f, 0.f), c1 = c0; // out-of-range texels read as 0
		if (aligned && x >= 0 && x + 1 < width)
		{
			const uint4 v = *reinterpret_cast<const uint4 *>(scene + row + x);
			c0 = bright_pass(unpack_half4(v.x, v.y));
			c1 = bright_pass(unpack_half4(v.z, v.w));
		}
		else
		{
			if (x >= 0 && x < width) c0 = bright_pass(load_half4(scene, row + x));
			if (x + 1 >= 0 && x + 1 < width) c1 = bright_pass(load_half4(scene, row + x + 1));
		}
		if (is_lit(c0) | is_lit(c1)) chunk_lit[i / 64] = 1; // texels i, i + 1 are in one chunk (i is even)
		even[i / 2] = make_float4(c0.x, c0.y, c0.z, c0.w);
		odd[i / 2] = make_float4(c1.x, c1.y, c1.z, c1.w);
	}
	__syncthreads();
	const int x = x0 + 2 * (int)threadIdx.x;
	// this wave's outputs are texels [128 w + 32, 128 w + 160) of the staged row; their windows reach 32 to either side
	const int w = threadIdx.x >> 6;
	const int wave_lit = chunk_lit[2 * w] | chunk_lit[2 * w + 1] | chunk_lit[2 * w + 2];
	vec4 s0 = V4(0.f, 0.f, 0.f, 0.f), s1 = s0;
	if (wave_lit && x < width) // else: 33 x (+0 + +-0 * w) = +0
	{
		Taps<-16, 1>::run(s0, &even[threadIdx.x + POST_HALO / 2]);
		Taps<-16, 1>::run(s1, &odd[threadIdx.x + POST_HALO / 2]);
	}
	const uint2 p0 = pack_half4(s0 * 2.f), p1 = pack_half4(s1 * 2.f);
	if (aligned && x + 1 < width)
		*reinterpret_cast<uint4 *>(bloom1 + row + x) = make_uint4(p0.x, p0.y, p1.x, p1.y);
	else
	{
		if (x < width) bloom1[row + x] = p0;
		if (x + 1 < width) bloom1[row + x + 1] = p1;
	}
	// a sum of non-zero terms may still round to zero halves: the flag describes what was STORED
	const bool mine = wave_lit && ((x < width && ((p0.x | p0.y) & 0x7fff7fffu) != 0u) || (x + 1 < width && ((p1.x | p1.y) & 0x7fff7fffu) != 0u));
	const unsigned long long stored = __ballot(mine);
	if ((threadIdx.x & 63) == 0) flags[(size_t)y * (gridDim.x * (POST_SEG / POST_FLAG_PIXELS)) + blockIdx.x * (POST_SEG / POST_FLAG_PIXELS) + w] = stored ? 1 : 0;
}

// ---- pass 2: vertical blur + tone map ---------------------------------------------------------------
#define POST_TX 32
#define POST_TY 32
// D3D's float -> UNORM8: saturate (NaN -> 0), scale, add a half, truncate.  After sat1 the value lies in
// [0.5, 255.5] and is never NaN, so the plain conversion (v_cvt_i32_f32: truncation) is all ftoi needs here.
__device__ __forceinline__ uint32_t to_unorm8(float v) { return (uint32_t)(int)(sat1(v) * 255.f + 0.5f); }
__device__ __forceinline__ float exp_d3d(float x) { return exp21(x * 1.44269504088896340736f); }
// pshader_hdr.hlsl:16-26 for one pixel: scene + bloom -> R8G8B8A8_UNORM
__device__ __forceinline__ uint32_t tone_map(vec4 sc, vec4 bloom)
{
	const vec4 total = sc + bloom;
	const vec4 e = -total * 1.f; // exposure 1
	const vec4 l = 1.f - V4(exp_d3d(e.x), exp_d3d(e.y), exp_d3d(e.z), exp_d3d(e.w));
	const float a = sc.w;
	const vec4 o = V4(lerp1(sc.x, l.x, a), lerp1(sc.y, l.y, a), lerp1(sc.z, l.z, a), lerp1(sc.w, l.w, a));
	return to_unorm8(o.x) | (to_unorm8(o.y) << 8) | (to_unorm8(o.z) << 16) | (to_unorm8(o.w) << 24);
}

__global__ __launch_bounds__(256) void k_bloom_v_tone(const uint2 *__restrict__ scene, const uint2 *__restrict__ bloom1,
	const unsigned char *__restrict__ flags, int flags_per_row, uint32_t *__restrict__ ldr, int width, int height)
{
	__shared__ float4 tile[POST_TY + 2 * POST_HALO][POST_TX];
	const int x0 = blockIdx.x * POST_TX, y0 = blockIdx.y * POST_TY;
	// Is any of the 96 rows of bloom texels this tile blurs over lit?  (A 32-pixel tile lies inside one flag's 128 pixels.)
	int row_lit = 0;
	if (threadIdx.x < POST_TY + 2 * POST_HALO)
	{
		const int y = y0 - POST_HALO + (int)threadIdx.x;
		if (y >= 0 && y < height) row_lit = flags[(size_t)y * flags_per_row + x0 / POST_FLAG_PIXELS];
	}
	if (!__syncthreads_or(row_lit))
	{
		// Dark tile: the blur is +0 everywhere, bloom2 = f16(+0 * 2) = +0; only the tone map is left.  No bloom
		// texel is read; a thread takes four neighbouring pixels of a row (32-byte loads, 16-byte stores).
		const int r = threadIdx.x >> 3, y = y0 + r, x = x0 + (threadIdx.x & 7) * 4;
		if (y >= height || x >= width) return;
		const vec4 dark = through_half4(V4(0.f, 0.f, 0.f, 0.f) * 2.f);
		const size_t at = (size_t)y * width + x;
		if (x + 3 < width && (at & 1) == 0) // 16-byte aligned: two texels per load
		{
			const uint4 s01 = *reinterpret_cast<const uint4 *>(scene + at), s23 = *reinterpret_cast<const uint4 *>(scene + at + 2);
			uint4 o;
			o.x = tone_map(unpack_half4(s01.x, s01.y), dark);
			o.y = tone_map(unpack_half4(s01.z, s01.w), dark);
			o.z = tone_map(unpack_half4(s23.x, s23.y), dark);
			o.w = tone_map(unpack_half4(s23.z, s23.w), dark);
			if ((at & 3) == 0)
				*reinterpret_cast<uint4 *>(ldr + at) = o;
			else
			{
				ldr[at] = o.x; ldr[at + 1] = o.y; ldr[at + 2] = o.z; ldr[at + 3] = o.w;
			}
		}
		else
		{
			for (int k = 0; k < 4 && x + k < width; ++k) ldr[at + k] = tone_map(load_half4(scene, at + k), dark);
		}
		return;
	}
	const int tx = threadIdx.x & (POST_TX - 1), ty = threadIdx.x / POST_TX; // 32 x 8
	const int x = x0 + tx;
	int lit = 0;
	for (int r = ty; r < POST_TY + 2 * POST_HALO; r += 8)
	{
		const int y = y0 - POST_HALO + r;
		vec4 c = V4(0.f, 0.f, 0.f, 0.f);
		if (x < width && y >= 0 && y < height) c = load_half4(bloom1, (size_t)y * width + x);
		lit |= is_lit(c);
		tile[r][tx] = make_float4(c.x, c.y, c.z, c.w);
	}
	const int block_lit = __syncthreads_or(lit); // the 128 pixels of the flag may be lit elsewhere than in these 32 columns
	if (x >= width) return;
	for (int r = ty; r < POST_TY; r += 8)
	{
		const int y = y0 + r;
		if (y >= height) break;
		vec4 sum = V4(0.f, 0.f, 0.f, 0.f);
		if (block_lit) Taps<-16, 2 * POST_TX>::run(sum, &tile[r + POST_HALO][tx]);
		const vec4 bloom = through_half4(sum * 2.f); // the reference stores bloom2 as f16
		ldr[(size_t)y * width + x] = tone_map(load_half4(scene, (size_t)y * width + x), bloom);
	}
}

size_t postprocess_flag_bytes(int width, int height) { return (size_t)((width + POST_SEG - 1) / POST_SEG) * (POST_SEG / POST_FLAG_PIXELS) * (size_t)height; }

hipError_t launch_postprocess(int width, int height, const void *scene16, void *bloom1, void *ldr8, unsigned char *flags, hipStream_t stream,
	hipEvent_t mid_event)
{
	const int segs = (width + POST_SEG - 1) / POST_SEG;
	dim3 g1(segs, height);
	hipLaunchKernelGGL(k_bloom_h, g1, dim3(POST_SEG / 2), 0, stream, reinterpret_cast<const uint2 *>(scene16), reinterpret_cast<uint2 *>(bloom1), flags, width, height);
	if (mid_event) (void)hipEventRecord(mid_event, stream);
	dim3 g2((width + POST_TX - 1) / POST_TX, (height + POST_TY - 1) / POST_TY);
	hipLaunchKernelGGL(k_bloom_v_tone, g2, dim3(256), 0, stream, reinterpret_cast<const uint2 *>(scene16), reinterpret_cast<const uint2 *>(bloom1),
		flags, segs * (POST_SEG / POST_FLAG_PIXELS), reinterpret_cast<uint32_t *>(ldr8), width, height);
	return hipGetLastError();
}

} // namespace sdfr
