// sdfr_post.hip -- the consumer of the raymarch output on gfx950: bloom + tone map.
//
// Mirrors the reference's HDR::process (Engine/Postprocessing.cpp:130-174): bright-pass +
// 33-tap stride-2 horizontal Gaussian (Engine/shader/bloom.hlsl:14-27), 33-tap stride-2
// vertical Gaussian (:29-38), exponential tone map blended by the alpha flag
// (Engine/shader/pshader_hdr.hlsl:16-26) into an R8G8B8A8_UNORM image.  The reference runs
// three passes over three RGBA16F textures; here two kernels:
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
// windows reach into holds light.  flags[y * n + x / POST_FLAG_PIXELS] (one byte per 32 pixels of a row: the width of a tile of pass 2)
// says whether anything but zeros was stored there: the vertical pass reads it instead of the bloom texels
// wherever it is clear.
#define POST_CHUNKS ((POST_SEG + 2 * POST_HALO) / 64)
#define POST_FLAG_PIXELS 32
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
	// 16 lanes = 32 pixels = one flag; a wave stores its four flags as one word
	const unsigned long long stored = __ballot(mine);
	if ((threadIdx.x & 63) == 0)
	{
		const uint32_t four = ((stored & 0xffffull) ? 1u : 0u) | ((stored & 0xffff0000ull) ? 0x100u : 0u) | ((stored & 0xffff00000000ull) ? 0x10000u : 0u) |
			((stored & 0xffff000000000000ull) ? 0x1000000u : 0u);
		*reinterpret_cast<uint32_t *>(flags + (size_t)y * (gridDim.x * (POST_SEG / POST_FLAG_PIXELS)) + blockIdx.x * (POST_SEG / POST_FLAG_PIXELS) + 4 * w) = four;
	}
}

// ---- pass 2: vertical blur + tone map ---------------------------------------------------------------
#ifndef POST_TX
#define POST_TX 16
#endif
#define POST_ROWS_PER_STEP (256 / POST_TX) // rows the block's 256 threads cover at once
#define POST_DARK_PIXELS (POST_TX * POST_TY / 256) // neighbouring pixels of a row one thread tone-maps in a dark tile: 2 or 4
#define POST_TY 32
// D3D's float -> UNORM8: saturate (NaN -> 0), scale, add a half, truncate.  After sat1 the value lies in
// [0.5, 255.5] and is never NaN, so the plain conversion (v_cvt_i32_f32: truncation) is all ftoi needs here.
__device__ __forceinline__ uint32_t to_unorm8(float v) { return (uint32_t)(int)(sat1(v) * 255.f + 0.5f); }
__device__ __forceinline__ float exp_d3d(float x) { return exp21(x * 1.44269504088896340736f); }
// exp21 for -126 <= y <= 0, where its guards (NaN, overflow, underflow, the 2^128 step) do nothing: the same
// operations in the same order, 13 instructions instead of 24 and no branches
__device__ __forceinline__ bool exp2_plain_range(float y) { return (y >= -126.f) & (y <= 0.f); } // NaN: false
__device__ __forceinline__ float exp2_plain(float y)
{
	const float k = rne1(y);
	const float f = y - k;
	float p = 1.535336188319500e-4f;
	p = fma1(p, f, 1.339887440266574e-3f);
	p = fma1(p, f, 9.618437357674640e-3f);
	p = fma1(p, f, 5.550332471162809e-2f);
	p = fma1(p, f, 2.402264791363012e-1f);
	p = fma1(p, f, 6.931472028550421e-1f);
	const float res = fma1(p, f, 1.0f);
	return res * bits_f32((uint32_t)((int)k + 127) << 23);
}
// one channel of pshader_hdr.hlsl:16-26 from its exponential on
__device__ __forceinline__ uint32_t tone_channel(float sc, float exponential, float a) { return to_unorm8(lerp1(sc, 1.f - exponential, a)); }
// pshader_hdr.hlsl:16-26 for one pixel: scene + bloom -> R8G8B8A8_UNORM.
//  * DARK: the caller knows bloom is +0 in all four channels.  scene + +0 differs from scene only for scene = -0
//    (the sum is +0), and the exponential of either zero is 1: the addition is left out.
//  * The exponentials: almost every texel has all four arguments in [-126, 0] (colours are not negative and below
//    87); if that holds for every lane of the wave, the guard-free form runs, else exp21 for all.
//  * Alpha: the renderer writes 0 or 1 there (|hdr flag|, sdfr_render_pixel.h), and without bloom on the alpha
//    channel the output byte is then a constant (0 -> 0; 1 -> the byte of lerp(1, 1 - exp(-1), 1)): if that holds for
//    every lane, the fourth exponential, blend and conversion are not computed.  Any other alpha takes the formula.
template <bool DARK>
__device__ __forceinline__ uint32_t tone_map(vec4 sc, vec4 bloom)
{
	const vec4 total = DARK ? sc : sc + bloom;
	const vec4 y = (-total * 1.f) * 1.44269504088896340736f; // exposure 1; exp(x) = exp2(x * log2(e)) as D3D compiles it
	const float a = sc.w;
	const bool flag_alpha = (DARK | (bloom.w == 0.f)) & ((sc.w == 0.f) | (sc.w == 1.f));
	uint32_t out;
	if (__all(exp2_plain_range(y.x) & exp2_plain_range(y.y) & exp2_plain_range(y.z) & (flag_alpha | exp2_plain_range(y.w))))
		out = tone_channel(sc.x, exp2_plain(y.x), a) | (tone_channel(sc.y, exp2_plain(y.y), a) << 8) | (tone_channel(sc.z, exp2_plain(y.z), a) << 16);
	else
		out = tone_channel(sc.x, exp21(y.x), a) | (tone_channel(sc.y, exp21(y.y), a) << 8) | (tone_channel(sc.z, exp21(y.z), a) << 16);
	if (__all(flag_alpha))
	{
		const uint32_t one = tone_channel(1.f, exp2_plain((-1.f * 1.f) * 1.44269504088896340736f), 1.f); // folds to a constant
		out |= (sc.w == 1.f ? one : 0u) << 24;
	}
	else
		out |= tone_channel(sc.w, exp21(y.w), a) << 24;
	return out;
}

// One block per 32 x 32 tile.  Tried and dropped: tone-mapping the dark tiles in a kernel of their own (no LDS, hence
// full residency: a black 4K frame in 24 instead of 31 us) and handing the lit ones to resident blocks through lists --
// the list-driven blocks needed 1.5x as long per lit tile as one block per tile does (dense frame 240 against 150 us,
// cube_sea 76 against 58 us), whatever the grid, the hand-out (strided, atomic cursor) or the tile-to-XCD mapping.
__global__ __launch_bounds__(256) void k_bloom_v_tone(const uint2 *__restrict__ scene, const uint2 *__restrict__ bloom1,
	const unsigned char *__restrict__ flags, int flags_per_row, uint32_t *__restrict__ ldr, int width, int height)
{
	__shared__ float4 tile[POST_TY + 2 * POST_HALO][POST_TX];
	__shared__ uint32_t rows_lit[3]; // which of the 96 staged rows hold light: bit r of rows_lit[r / 32]
	const int x0 = blockIdx.x * POST_TX, y0 = blockIdx.y * POST_TY;
	// A dark tile (below) only tone-maps: a thread takes POST_DARK_PIXELS neighbouring pixels of a row (16-byte loads).
	// Their loads are issued before the flags are looked at, so that the two round trips overlap; a lit tile drops
	// them (it is the rarer case and reads its scene texels in another arrangement).
	const int dr = threadIdx.x >> 3, dy = y0 + dr, dx = x0 + (threadIdx.x & 7) * POST_DARK_PIXELS;
	const size_t at = (size_t)dy * width + dx;
	const bool mine = dy < height && dx < width;
	const bool wide = mine && dx + POST_DARK_PIXELS - 1 < width && (at & 1) == 0; // 16-byte aligned: two texels per load
	uint4 s01 = make_uint4(0u, 0u, 0u, 0u), s23 = s01;
	if (wide)
	{
		s01 = *reinterpret_cast<const uint4 *>(scene + at);
		if (POST_DARK_PIXELS == 4) s23 = *reinterpret_cast<const uint4 *>(scene + at + 2);
	}
	// Is any of the 96 rows of bloom texels this tile blurs over lit?  (A tile is no wider than a flag.)
	int row_lit = 0;
	if (threadIdx.x < POST_TY + 2 * POST_HALO)
	{
		const int y = y0 - POST_HALO + (int)threadIdx.x;
		if (y >= 0 && y < height) row_lit = flags[(size_t)y * flags_per_row + x0 / POST_FLAG_PIXELS];
	}
	if (threadIdx.x < 3) rows_lit[threadIdx.x] = 0u;
	if (!__syncthreads_or(row_lit))
	{
		// Dark tile: the blur is +0 everywhere, bloom2 = f16(+0 * 2) = +0; only the tone map is left.  No bloom texel is read.
		if (!mine) return;
		const vec4 dark = through_half4(V4(0.f, 0.f, 0.f, 0.f) * 2.f);
		if (wide)
		{
			uint4 o;
			o.x = tone_map<true>(unpack_half4(s01.x, s01.y), dark);
			o.y = tone_map<true>(unpack_half4(s01.z, s01.w), dark);
			if (POST_DARK_PIXELS == 4)
			{
				o.z = tone_map<true>(unpack_half4(s23.x, s23.y), dark);
				o.w = tone_map<true>(unpack_half4(s23.z, s23.w), dark);
				if ((at & 3) == 0)
					*reinterpret_cast<uint4 *>(ldr + at) = o;
				else
				{
					ldr[at] = o.x; ldr[at + 1] = o.y; ldr[at + 2] = o.z; ldr[at + 3] = o.w;
				}
			}
			else
				*reinterpret_cast<uint2 *>(ldr + at) = make_uint2(o.x, o.y); // at is even: 8-byte aligned
		}
		else
		{
			for (int k = 0; k < POST_DARK_PIXELS && dx + k < width; ++k) ldr[at + k] = tone_map<true>(load_half4(scene, at + k), dark);
		}
		return;
	}
	const int tx = threadIdx.x & (POST_TX - 1), ty = threadIdx.x / POST_TX;
	const int x = x0 + tx;
	for (int r = ty; r < POST_TY + 2 * POST_HALO; r += POST_ROWS_PER_STEP)
	{
		const int y = y0 - POST_HALO + r;
		vec4 c = V4(0.f, 0.f, 0.f, 0.f);
		if (x < width && y >= 0 && y < height) c = load_half4(bloom1, (size_t)y * width + x);
		tile[r][tx] = make_float4(c.x, c.y, c.z, c.w);
		// a wave stages 64 / POST_TX rows (POST_TX lanes each), the first of them -- lane 0's r -- a multiple of that
		const unsigned long long lit = __ballot(is_lit(c));
		if ((threadIdx.x & 63) == 0 && lit)
		{
			uint32_t rows = 0u;
			for (int g = 0; g < 64 / POST_TX; ++g) rows |= ((lit >> (g * POST_TX)) & ((1ull << POST_TX) - 1ull)) ? 1u << g : 0u;
			atomicOr(&rows_lit[r >> 5], rows << (r & 31));
		}
	}
	__syncthreads();
	if (x >= width) return;
	const unsigned long long low = (unsigned long long)rows_lit[0] | ((unsigned long long)rows_lit[1] << 32);
	const unsigned long long high = rows_lit[2];
	for (int r = ty; r < POST_TY; r += POST_ROWS_PER_STEP)
	{
		const int y = y0 + r;
		if (y >= height) break;
		vec4 sum = V4(0.f, 0.f, 0.f, 0.f);
		// output row r blurs staged rows r .. r + 64: rows r .. 63 of `low` and rows 64 .. 64 + r of `high`; nothing there: a sum of zeros
		if (((low >> r) | (high & ((2ull << r) - 1ull))) != 0ull) Taps<-16, 2 * POST_TX>::run(sum, &tile[r + POST_HALO][tx]);
		const vec4 bloom = through_half4(sum * 2.f); // the reference stores bloom2 as f16
		ldr[(size_t)y * width + x] = tone_map<false>(load_half4(scene, (size_t)y * width + x), bloom);
	}
}

// the scratch of a frame: one flag per 32 pixels of a row
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
