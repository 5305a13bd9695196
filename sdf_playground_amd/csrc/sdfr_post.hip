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
// to the tone map (pass 2) -- the same bytes, at the speed of the memory system.  Blocks that do see light
// run the taps as before.
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

// ---- pass 1: bright-pass + horizontal blur ---------------------------------------------------------
__global__ __launch_bounds__(256) void k_bloom_h(const uint2 *__restrict__ scene, uint2 *__restrict__ bloom1, int width, int height)
{
	__shared__ float4 tile[256 + 2 * POST_HALO];
	const int y = blockIdx.y;
	const int x0 = blockIdx.x * 256;
	int lit = 0; // does this thread stage anything but zeros (of either sign)?
	for (int i = threadIdx.x; i < 256 + 2 * POST_HALO; i += 256)
	{
		const int x = x0 - POST_HALO + i;
		vec4 c = V4(0.f, 0.f, 0.f, 0.f); // out-of-range texels read as 0
		if (x >= 0 && x < width)
		{
			c = load_half4(scene, (size_t)y * width + x);
			const float brightness = dot(V3(c.x, c.y, c.z), V3(0.2126f, 0.7152f, 0.0722f));
			const float factor = sat1((sat1(brightness) - 0.75f) * 4.f);
			c = c * factor;
		}
		lit |= !(c.x == 0.f && c.y == 0.f && c.z == 0.f && c.w == 0.f); // NaN counts as lit
		tile[i] = make_float4(c.x, c.y, c.z, c.w);
	}
	const int block_lit = __syncthreads_or(lit);
	const int x = x0 + threadIdx.x;
	if (x >= width) return;
	vec4 sum = V4(0.f, 0.f, 0.f, 0.f);
	if (block_lit) Taps<-16, 2>::run(sum, &tile[threadIdx.x + POST_HALO]); // else: 33 x (+0 + +-0 * w) = +0
	bloom1[(size_t)y * width + x] = pack_half4(sum * 2.f);
}

// ---- pass 2: vertical blur + tone map ---------------------------------------------------------------
#define POST_TX 32
#define POST_TY 32
__device__ __forceinline__ uint32_t to_unorm8(float v) { return (uint32_t)ftoi1(sat1(v) * 255.f + 0.5f); }
__device__ __forceinline__ float exp_d3d(float x) { return exp21(x * 1.44269504088896340736f); }

__global__ __launch_bounds__(256) void k_bloom_v_tone(const uint2 *__restrict__ scene, const uint2 *__restrict__ bloom1, uint32_t *__restrict__ ldr,
	int width, int height)
{
	__shared__ float4 tile[POST_TY + 2 * POST_HALO][POST_TX];
	const int tx = threadIdx.x & (POST_TX - 1), ty = threadIdx.x / POST_TX; // 32 x 8
	const int x = blockIdx.x * POST_TX + tx;
	const int y0 = blockIdx.y * POST_TY;
	int lit = 0;
	for (int r = ty; r < POST_TY + 2 * POST_HALO; r += 8)
	{
		const int y = y0 - POST_HALO + r;
		vec4 c = V4(0.f, 0.f, 0.f, 0.f);
		if (x < width && y >= 0 && y < height) c = load_half4(bloom1, (size_t)y * width + x);
		lit |= !(c.x == 0.f && c.y == 0.f && c.z == 0.f && c.w == 0.f);
		tile[r][tx] = make_float4(c.x, c.y, c.z, c.w);
	}
	const int block_lit = __syncthreads_or(lit);
	if (x >= width) return;
	for (int r = ty; r < POST_TY; r += 8)
	{
		const int y = y0 + r;
		if (y >= height) break;
		vec4 sum = V4(0.f, 0.f, 0.f, 0.f);
		if (block_lit) Taps<-16, 2 * POST_TX>::run(sum, &tile[r + POST_HALO][tx]);
		const vec4 bloom = through_half4(sum * 2.f); // the reference stores bloom2 as f16
		const vec4 sc = load_half4(scene, (size_t)y * width + x);
		const vec4 total = sc + bloom;
		const vec4 e = -total * 1.f; // exposure 1
		const vec4 l = 1.f - V4(exp_d3d(e.x), exp_d3d(e.y), exp_d3d(e.z), exp_d3d(e.w));
		const float a = sc.w;
		const vec4 o = V4(lerp1(sc.x, l.x, a), lerp1(sc.y, l.y, a), lerp1(sc.z, l.z, a), lerp1(sc.w, l.w, a));
		ldr[(size_t)y * width + x] = to_unorm8(o.x) | (to_unorm8(o.y) << 8) | (to_unorm8(o.z) << 16) | (to_unorm8(o.w) << 24);
	}
}

hipError_t launch_postprocess(int width, int height, const void *scene16, void *bloom1, void *ldr8, hipStream_t stream, hipEvent_t mid_event)
{
	dim3 g1((width + 255) / 256, height);
	hipLaunchKernelGGL(k_bloom_h, g1, dim3(256), 0, stream, reinterpret_cast<const uint2 *>(scene16), reinterpret_cast<uint2 *>(bloom1), width, height);
	if (mid_event) (void)hipEventRecord(mid_event, stream);
	dim3 g2((width + POST_TX - 1) / POST_TX, (height + POST_TY - 1) / POST_TY);
	hipLaunchKernelGGL(k_bloom_v_tone, g2, dim3(256), 0, stream, reinterpret_cast<const uint2 *>(scene16), reinterpret_cast<const uint2 *>(bloom1),
		reinterpret_cast<uint32_t *>(ldr8), width, height);
	return hipGetLastError();
}

} // namespace sdfr
