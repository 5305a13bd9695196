// sdfr_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the SDF raymarch renderer.
//
// Two schedules of the same pipeline stages (sdfr_pixel.h); both produce identical pixels.
//
//  * WAVEFRONT.  The path is VALU-bound: what matters is keeping all
//    64 lanes of every wave busy although rays need 1..iter_count steps.  Per bounce round:
//      k_march  persistent waves; every lane sphere-traces ONE ray at a time and, when its ray
//               ends (hit: plus the 3 forward-difference normal samples, taken in the same
//               loop so the scene function stays convergent), goes idle; when a ballot shows
//               enough idle lanes the wave refills them from its current range of the round's
//               ray list.  Ray records live in HBM as structure-of-arrays, so a refill reads
//               consecutive addresses.  A wave claims ranges of 128 list entries from a
//               per-round cursor (one atomic per range), so the load balances itself
//               whatever the grid size and the real residency are.
//      k_shade  one lane per finished ray: material, lighting, secondary-ray spawn into the
//               pixel's 8-slot queue (HBM), pop of the pixel's next ray, append to the next
//               round's list with ONE atomic per block.
//    Rounds are bounded by bounce_count (16); the list sizes stay on the device, empty
//    rounds cost a few microseconds each.
//
//  * PIXEL (default: the faster one on every measured scene).  One lane per pixel
//    runs the reference's bounce loop start to finish (sdfr_render_pixel.h,
//    sdfr_pixel_kernel.h); waves cover 8x8 pixel tiles.
//
// Everything wave-uniform (camera, variables, limits, per-frame scene constants) travels as
// a by-value kernel argument and therefore sits in SGPRs.
#include "sdfr_kernels.h"
#include "sdfr_perpixel.h"
#include "sdfr_pixel_kernel.h"

namespace sdfr {

// refill a march wave once this many lanes are idle (or when all are)
#define SDFR_REFILL_THRESHOLD 16
// list entries a march wave claims per atomic
#define SDFR_GRAB 128

uint32_t launch_work_items(int width, const RowMap &rm)
{
	const uint32_t tw_log2 = (uint32_t)rm.tile_w_log2, th_log2 = 6u - tw_log2;
	const uint32_t tiles_x = ((uint32_t)width + (1u << tw_log2) - 1u) >> tw_log2;
	const uint32_t tiles_y = ((uint32_t)rm.local_rows + (1u << th_log2) - 1u) >> th_log2;
	return tiles_x * tiles_y * 64u;
}
static uint32_t work_items(const FrameU &U, const RowMap &rm) { return launch_work_items(U.width, rm); }

// =================================================================================================
// PIXEL schedule (body: sdfr_pixel_kernel.h)
// =================================================================================================
template <class Scene, bool DBG>
__global__ SDFR_PIXEL_KERNEL_ATTRS void k_pixel(FrameU U, RowMap rm, uint32_t n_work, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *partials, RenderTotals *totals, float *ray_queue, size_t cap)
{
	pixel_kernel<Scene, DBG>(U, rm, n_work, out, format, pixel_stats, partials, totals, ray_queue, cap);
}

// Folds the per-block partial sums of a pixel-schedule launch (one 32-byte record per wave: 4 MB
// at 4K) into the render totals, which the pixel kernel has cleared: a few blocks, each keeping
// several independent loads in flight per thread, then ONE set of four atomics per block.
#define SDFR_REDUCE_THREADS 256
#define SDFR_REDUCE_BLOCKS 64
__global__ __launch_bounds__(SDFR_REDUCE_THREADS) void k_reduce_totals(const RenderTotals *__restrict__ partials, uint32_t n, RenderTotals *totals)
{
	__shared__ unsigned long long acc[4];
	if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
	__syncthreads();
	unsigned long long s[4] = {0ull, 0ull, 0ull, 0ull};
	const ulonglong4 *src = reinterpret_cast<const ulonglong4 *>(partials);
	const uint32_t stride = gridDim.x * SDFR_REDUCE_THREADS;
	for (uint32_t base = blockIdx.x * SDFR_REDUCE_THREADS + threadIdx.x; base < n; base += stride * 4)
	{
		ulonglong4 v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			const uint32_t i = base + (uint32_t)k * stride;
			v[k] = i < n ? src[i] : make_ulonglong4(0ull, 0ull, 0ull, 0ull);
		}
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			s[0] += v[k].x;
			s[1] += v[k].y;
			s[2] += v[k].z;
			s[3] += v[k].w;
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		for (int k = 0; k < 4; ++k) s[k] += __shfl_down(s[k], off);
	if ((threadIdx.x & 63) == 0)
		for (int k = 0; k < 4; ++k) atomicAdd(&acc[k], s[k]);
	__syncthreads();
	if (threadIdx.x < 4 && acc[threadIdx.x]) atomicAdd(reinterpret_cast<unsigned long long *>(totals) + threadIdx.x, acc[threadIdx.x]);
}
hipError_t launch_reduce_totals(const RenderTotals *partials, uint32_t n_blocks, RenderTotals *totals, hipStream_t stream)
{
	uint32_t blocks = (n_blocks + SDFR_REDUCE_THREADS * 4 - 1) / (SDFR_REDUCE_THREADS * 4);
	if (blocks > SDFR_REDUCE_BLOCKS) blocks = SDFR_REDUCE_BLOCKS;
	if (blocks < 1) blocks = 1;
	hipLaunchKernelGGL(k_reduce_totals, dim3(blocks), dim3(SDFR_REDUCE_THREADS), 0, stream, partials, n_blocks, totals);
	return hipGetLastError();
}

// =================================================================================================
// WAVEFRONT schedule
// =================================================================================================
// march result fields
enum { RS_STATUS = 0, RS_T, RS_D, RS_NX, RS_NY, RS_NZ, RS_COUNT };
// counters[]: [r] = size of round r's list (r = 0..16); [32 + r] = march cursor of round r
enum { CNT_ROUND0 = 0 };


// ---- k_init: primary rays, empty queues, cleared accumulators, round-0 list ---------------------
__global__ __launch_bounds__(SDFR_BLOCK) void k_init(FrameU U, RowMap rm, uint32_t n_work, WavefrontWorkspace ws, uint32_t *pixel_stats)
{
	const uint32_t w = blockIdx.x * SDFR_BLOCK + threadIdx.x;
	if (w == 0)
	{
		ws.counters[CNT_ROUND0] = n_work;
		for (int r = 1; r <= 16; ++r) ws.counters[r] = 0;
		for (int r = 0; r < 16; ++r) ws.counters[32 + r] = 0; // march cursors
	}
	if (w >= n_work) return;
	PixelCoord pc;
	if (!work_to_pixel(U, rm, w, pc))
	{
		ws.list_a[w] = SDFR_INVALID_PIXEL;
		return;
	}
	const size_t cap = ws.capacity;
	const PixelRay pr = pixel_ray(U, pc.px, pc.py);
	store_ray(ws.ray_cur, cap, pc.pid, primary_ray(U, pr));
	ws.qdepth_lo[pc.pid] = 0xffffffffu;
	ws.qdepth_hi[pc.pid] = 0xffffffffu;
	ws.accum[0 * cap + pc.pid] = 0.f;
	ws.accum[1 * cap + pc.pid] = 0.f;
	ws.accum[2 * cap + pc.pid] = 0.f;
	ws.accum[3 * cap + pc.pid] = -1.f; // hdr_output "not set" (pshader_sdf.hlsl:284)
	ws.list_a[w] = pc.pid;
	if (pixel_stats)
	{
		pixel_stats[3 * (size_t)pc.pid + 0] = 0;
		pixel_stats[3 * (size_t)pc.pid + 1] = 0;
		pixel_stats[3 * (size_t)pc.pid + 2] = 0;
	}
}

// ---- k_march ----------------------------------------------------------------------------------------
enum { LANE_IDLE = 0, LANE_MARCH = 1, LANE_GRAD0 = 2, LANE_GRAD1 = 3, LANE_GRAD2 = 4 };

template <class Scene, bool DBG>
__global__ __launch_bounds__(SDFR_BLOCK) void k_march(FrameU U, WavefrontWorkspace ws, const uint32_t *__restrict__ list,
	const uint32_t *__restrict__ n_ptr, uint32_t *cursor, uint32_t *pixel_stats, RenderTotals *totals)
{
	const uint32_t n = *n_ptr;
	const size_t cap = ws.capacity;
	const DebugFlags F = debug_flags(U);
	const uint32_t lane = threadIdx.x & 63u;
	// [next, end) = unconsumed part of the list range this wave currently owns (wave-uniform).
	// Ranges of SDFR_GRAB entries are claimed from a per-round cursor with one atomic each, so
	// the load balances itself whatever the grid size and residency are.
	uint32_t next = 0, end = 0;
	bool exhausted = n == 0;

	int state = LANE_IDLE;
	uint32_t pid = 0;
	March m = march_begin(V3s(0.f), V3s(0.f));
	typename Scene::RayInv R = {};
	float inside_sign = 1.f, max_range = 0.f, baseline = 0.f, g0 = 0.f, g1 = 0.f;
	uint32_t evals = 0;       // of the current ray
	uint32_t tot_evals = 0, tot_hits = 0;

	for (;;)
	{
		const unsigned long long idle = __ballot(state == LANE_IDLE);
		if (idle)
		{
			if (next == end && !exhausted)
			{
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd(cursor, (uint32_t)SDFR_GRAB);
				base = __builtin_amdgcn_readfirstlane(base);
				if (base >= n)
					exhausted = true;
				else
				{
					next = base;
					end = base + SDFR_GRAB < n ? base + SDFR_GRAB : n;
				}
			}
			const uint32_t n_idle = (uint32_t)__popcll(idle);
			if (next == end)
			{
				if (n_idle == 64u) break; // list drained and every lane finished
			}
			else if (n_idle >= SDFR_REFILL_THRESHOLD || n_idle == 64u)
			{
				// hand the next list entries to the idle lanes, in lane order
				const uint32_t avail = end - next;
				if (state == LANE_IDLE)
				{
					const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
					if (rank < avail)
					{
						const uint32_t p = list[next + rank];
						if (p != SDFR_INVALID_PIXEL)
						{
							pid = p;
							const RayRec ray = load_ray(ws.ray_cur, cap, pid);
							R = Scene::ray_setup(U, ray.dir, ray_flags(ray));
							inside_sign = ray_inside_sign(ray);
							max_range = ray_is_shadow(ray) ? ray.shadow_range : U.range;
							m = march_begin(ray.pos, ray.dir);
							evals = 0;
							state = LANE_MARCH;
						}
					}
				}
				next += n_idle < avail ? n_idle : avail;
			}
		}

		if (state != LANE_IDLE)
		{
			// one scene-distance evaluation per lane: a march sample or a normal sample
			const bool marching = state == LANE_MARCH;
			if (marching) march_pre(m);
			const vec3 hp = march_pos(m);
			vec3 p = hp;
			if (state == LANE_GRAD0) p = grad_sample_pos(hp, 0, SDFR_GRAD_EPS);
			if (state == LANE_GRAD1) p = grad_sample_pos(hp, 1, SDFR_GRAD_EPS);
			if (state == LANE_GRAD2) p = grad_sample_pos(hp, 2, SDFR_GRAD_EPS);
			const float dist = map_geometry<Scene, DBG>(U, F, R, p, m.dir, marching);

			if (marching)
			{
				evals++;
				const int status = march_advance(m, dist * inside_sign, max_range, (uint32_t)U.iter_count);
				if (status == MARCH_HIT)
				{
					baseline = m.d * inside_sign;
					state = LANE_GRAD0;
				}
				else if (status == MARCH_MISS)
				{
					ws.result[RS_STATUS * cap + pid] = __uint_as_float((uint32_t)MARCH_MISS << 24 | m.iter);
					tot_evals += evals;
					if (pixel_stats) pixel_stats[3 * (size_t)pid + 1] += evals;
					state = LANE_IDLE;
				}
			}
			else if (state == LANE_GRAD0)
			{
				g0 = dist - baseline;
				state = LANE_GRAD1;
			}
			else if (state == LANE_GRAD1)
			{
				g1 = dist - baseline;
				state = LANE_GRAD2;
			}
			else
			{
				const vec3 nrm = normalize(V3(g0, g1, dist - baseline));
				ws.result[RS_STATUS * cap + pid] = __uint_as_float((uint32_t)MARCH_HIT << 24 | m.iter);
				ws.result[RS_T * cap + pid] = m.t;
				ws.result[RS_D * cap + pid] = m.d;
				ws.result[RS_NX * cap + pid] = nrm.x;
				ws.result[RS_NY * cap + pid] = nrm.y;
				ws.result[RS_NZ * cap + pid] = nrm.z;
				tot_evals += evals;
				tot_hits += 1;
				if (pixel_stats) pixel_stats[3 * (size_t)pid + 1] += evals;
				state = LANE_IDLE;
			}
		}
	}
	block_add_totals(totals, 0, 0, tot_evals, tot_hits);
}

// ---- k_shade -----------------------------------------------------------------------------------------
template <class Scene, bool DBG>
__global__ __launch_bounds__(SDFR_BLOCK) void k_shade(FrameU U, RowMap rm, WavefrontWorkspace ws, const uint32_t *__restrict__ list,
	const uint32_t *__restrict__ n_ptr, uint32_t *__restrict__ next_list, uint32_t *next_n, int round, void *out, int format,
	uint32_t *pixel_stats, RenderTotals *totals)
{
	const uint32_t n = *n_ptr;
	const size_t cap = ws.capacity;
	const DebugFlags F = debug_flags(U);
	__shared__ uint32_t s_count, s_base;
	uint32_t n_rays = 0, n_done = 0;

	for (uint32_t base = blockIdx.x * SDFR_BLOCK; base < n; base += gridDim.x * SDFR_BLOCK)
	{
		if (threadIdx.x == 0) s_count = 0;
		__syncthreads();
		const uint32_t i = base + threadIdx.x;
		uint32_t pid = SDFR_INVALID_PIXEL;
		if (i < n) pid = list[i];
		bool alive = false;
		if (pid != SDFR_INVALID_PIXEL)
		{
			n_rays++;
			const RayRec ray = load_ray(ws.ray_cur, cap, pid);
			int px, py;
			pid_to_pixel(U, rm, pid, px, py);
			const PixelRay pr = pixel_ray(U, px, py);
			const uint32_t status_iter = __float_as_uint(ws.result[RS_STATUS * cap + pid]);
			const uint32_t status = status_iter >> 24, iter = status_iter & 0xffffffu;

			uint64_t depths = (uint64_t)ws.qdepth_lo[pid] | ((uint64_t)ws.qdepth_hi[pid] << 32);
			int count = 0;
			for (int s = 0; s < SDFR_MAX_RAYS; ++s)
				count += (((depths >> (8 * s)) & 0xffu) != RAY_DEPTH_INVALID) ? 1 : 0;
			float hdr = ws.accum[3 * cap + pid];

			vec3 add;
			GlobalRayStore store = {ws.ray_queue, cap, pid}; // the pixel's pending rays (48-byte records)
			if (status == MARCH_HIT)
			{
				HitInfo hit;
				hit.t = ws.result[RS_T * cap + pid];
				hit.d = ws.result[RS_D * cap + pid];
				hit.iter = iter;
				hit.normal = V3(ws.result[RS_NX * cap + pid], ws.result[RS_NY * cap + pid], ws.result[RS_NZ * cap + pid]);
				hit.pos = mad(ray.dir, hit.t, ray.pos);
				const float max_range = ray_is_shadow(ray) ? ray.shadow_range : U.range;
				Spawner<GlobalRayStore> q(store, depths, count, U.ray_count);
				add = shade_hit<Scene, DBG, GlobalRayStore>(U, F, ray, pr, hit, max_range, hdr, q);
				depths = q.depths;
				count = q.count;
				if (pixel_stats) pixel_stats[3 * (size_t)pid + 2] += 1;
			}
			else
			{
				add = shade_miss<Scene>(U, ray, iter);
			}
			const vec3 acc = V3(ws.accum[0 * cap + pid], ws.accum[1 * cap + pid], ws.accum[2 * cap + pid]) + add;
			if (pixel_stats) pixel_stats[3 * (size_t)pid + 0] += 1;

			if (count > 0 && round + 1 < U.bounce_count)
			{
				// pop the pixel's next ray: it is traced in the next round
				const int slot = queue_next(depths, U.ray_count);
				const RayRec nr = store.get(slot);
				store_ray(ws.ray_cur, cap, pid, nr);
				depths = queue_set_depth(depths, slot, RAY_DEPTH_INVALID);
				ws.qdepth_lo[pid] = (uint32_t)depths;
				ws.qdepth_hi[pid] = (uint32_t)(depths >> 32);
				ws.accum[0 * cap + pid] = acc.x;
				ws.accum[1 * cap + pid] = acc.y;
				ws.accum[2 * cap + pid] = acc.z;
				ws.accum[3 * cap + pid] = hdr;
				alive = true;
			}
			else
			{
				store_pixel(out, format, pid, V4(acc.x, acc.y, acc.z, abs1(hdr)), (uint32_t)rm.local_rows * (uint32_t)U.width);
				n_done++;
			}
		}
		// append the surviving pixels to the next round's list: one atomic per block
		uint32_t my_off = 0;
		if (alive) my_off = atomicAdd(&s_count, 1u);
		__syncthreads();
		if (threadIdx.x == 0 && s_count) s_base = atomicAdd(next_n, s_count);
		__syncthreads();
		if (alive) next_list[s_base + my_off] = pid;
		__syncthreads();
	}
	block_add_totals(totals, n_done, n_rays, 0, 0);
}

// ---- strip assembly on the root (multi-GPU) ---------------------------------------------------------
// gathered[rank][local pixel] -> image[global pixel].  One block row per image row (blockIdx.y), so
// the strip arithmetic is wave-uniform and there is no per-pixel division; one 16-byte store per
// pixel.  RGBA32F / RGBA16F strips are copied; packed strips (12-byte float or 6-byte half rgb
// triples, then one flag byte per pixel, per rank) expand to RGBA32F / RGBA16F.
__global__ __launch_bounds__(SDFR_BLOCK) void k_assemble(int width, int height, int world, size_t strip_pixels, const uint32_t *gathered,
	uint32_t *image, int format, int priv_count, int priv_period)
{
	const uint32_t px = blockIdx.x * SDFR_BLOCK + threadIdx.x, py = blockIdx.y;
	if (px >= (uint32_t)width) return;
	uint32_t strip = py >> 3; // becomes the index among the shared strips
	if (priv_count > 0)
	{
		const uint32_t j = strip % (uint32_t)priv_period;
		if (j < (uint32_t)priv_count) return; // a private strip: rank 0 rendered it straight into the image
		strip = (strip / (uint32_t)priv_period) * (uint32_t)(priv_period - priv_count) + (j - (uint32_t)priv_count);
	}
	const uint32_t rank = strip % (uint32_t)world, local_strip = strip / (uint32_t)world;
	const size_t lpix = ((size_t)local_strip * 8u + (py & 7u)) * (size_t)width + px;
	const size_t g = (size_t)py * (size_t)width + px;
	if (format == FORMAT_STRIP_RGB32F_A8)
	{
		const size_t packed_rank_bytes = (13 * strip_pixels + 3) & ~(size_t)3;
		const unsigned char *base = reinterpret_cast<const unsigned char *>(gathered) + (size_t)rank * packed_rank_bytes;
		const uint32_t *rgb = reinterpret_cast<const uint32_t *>(base) + 3 * lpix;
		reinterpret_cast<uint4 *>(image)[g] = make_uint4(rgb[0], rgb[1], rgb[2], base[12 * strip_pixels + lpix] ? 0x3f800000u : 0u);
	}
	else if (format == FORMAT_STRIP_RGB16F_A8)
	{
		// 6-byte rgb half triples, then one flag byte per pixel, per rank -> RGBA16F (alpha 1.0 = 0x3c00)
		const size_t packed_rank_bytes = (7 * strip_pixels + 3) & ~(size_t)3;
		const unsigned char *base = reinterpret_cast<const unsigned char *>(gathered) + (size_t)rank * packed_rank_bytes;
		const unsigned short *rgb = reinterpret_cast<const unsigned short *>(base) + 3 * lpix;
		const uint32_t a = base[6 * strip_pixels + lpix] ? 0x3c00u : 0u;
		reinterpret_cast<uint2 *>(image)[g] = make_uint2((uint32_t)rgb[0] | ((uint32_t)rgb[1] << 16), (uint32_t)rgb[2] | (a << 16));
	}
	else if (format == FORMAT_RGBA32F)
	{
		reinterpret_cast<uint4 *>(image)[g] = reinterpret_cast<const uint4 *>(gathered)[(size_t)rank * strip_pixels + lpix];
	}
	else
	{
		reinterpret_cast<uint2 *>(image)[g] = reinterpret_cast<const uint2 *>(gathered)[(size_t)rank * strip_pixels + lpix];
	}
}

// ---- self-test of the fast exact arithmetic (sdfr_math.h: sqrt1, div_c) ------------------------------
// what = 0: sqrt1(a) against the generic IEEE lowering for a = +0 and every a in [2^-96, FLT_MAX]
// what = 1: div_c(a, c, 1/c) against a / c for a = +0 and every 2^-100 <= |a| <= 2^110
// what = 2: negative control -- the plain reciprocal multiply a * (1/c) on the same inputs (must differ)
// what = 3: div_c(a, c, 1/c) against a / c for a = +0 and every 2^-60 <= |a| <= 2^40 (the fast
//           ground plane: numerator = height above the floor, c = per-ray denominator in [1e-20, 2])
__global__ __launch_bounds__(SDFR_BLOCK) void k_selftest_math(int what, float c, unsigned long long *mismatches)
{
	const unsigned long long stride = (unsigned long long)gridDim.x * SDFR_BLOCK;
	unsigned int bad = 0;
	const float rc = 1.0f / c;
	for (unsigned long long u = (unsigned long long)blockIdx.x * SDFR_BLOCK + threadIdx.x; u < (1ull << 32); u += stride)
	{
		const float a = __uint_as_float((uint32_t)u);
		float ref, got;
		if (what == 0)
		{
			if (!(a == 0.f && u == 0) && !(a >= 0x1p-96f && a <= 3.402823466e+38f)) continue;
			ref = sqrt_ieee(a);
			got = sqrt1(a);
		}
		else if (what == 4 || what == 5)
		{
			// what = 4: rcp1(a) against the IEEE 1 / a for a = +-0, +-inf and 2^-100 <= |a| <= 2^100
			// what = 5: negative control -- the bare v_rcp_f32 (1 ulp) on the same inputs (must differ)
			const float m = abs1(a);
			if (!(m == 0.f) && !(m >= 0x1p-100f && m <= 0x1p100f) && !(m > 3.402823466e+38f)) continue;
			ref = 1.0f / a;
			got = what == 4 ? rcp1(a) : __builtin_amdgcn_rcpf(a);
		}
		else
		{
			const float m = abs1(a);
			if (what == 3)
			{
				if (!(m == 0.f) && !(m >= 0x1p-60f && m <= 0x1p40f)) continue;
			}
			else if (!(m == 0.f) && !(m >= 0x1p-100f && m <= 0x1p110f)) // up to 2^110: the smooth minima start from a 3e30 sentinel
				continue;
			ref = a / c;
			got = what == 2 ? a * rc : div_c(a, c, rc);
		}
		if (__float_as_uint(ref) != __float_as_uint(got)) bad++;
	}
	if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

hipError_t launch_selftest_math(int what, float c, unsigned long long *d_mismatches, hipStream_t stream)
{
	hipLaunchKernelGGL(k_selftest_math, dim3(8192), dim3(SDFR_BLOCK), 0, stream, what, c, d_mismatches);
	return hipGetLastError();
}

// =================================================================================================
// launchers
// =================================================================================================
int pixel_block_threads() { return SDFR_PIXEL_BLOCK; }

int device_cu_count(int device)
{
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 256;
	return prop.multiProcessorCount;
}

template <class Scene, bool DBG>
static hipError_t run_pixel(const FrameU &U, const RowMap &rm, void *out, int format, uint32_t *pixel_stats, RenderTotals *totals,
	const WavefrontWorkspace &ws, hipStream_t stream)
{
	const uint32_t n_work = work_items(U, rm);
	if ((size_t)n_work > ws.capacity) return hipErrorInvalidValue;
	const uint32_t blocks = (n_work + SDFR_PIXEL_BLOCK - 1) / SDFR_PIXEL_BLOCK;
	hipLaunchKernelGGL((k_pixel<Scene, DBG>), dim3(blocks), dim3(SDFR_PIXEL_BLOCK), 0, stream, U, rm, n_work, out, format, pixel_stats, ws.partials, totals,
		ws.ray_queue, ws.capacity);
	return launch_reduce_totals(ws.partials, blocks, totals, stream);
}

hipError_t launch_pixel_schedule(int scene, const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream)
{
	switch (scene)
	{
#define SDFR_RUN(I, S) case I: return frame_needs_debug(U) ? run_pixel<S, true>(U, rows, out, format, pixel_stats, totals, ws, stream) : run_pixel<S, false>(U, rows, out, format, pixel_stats, totals, ws, stream);
		SDFR_FOR_EACH_SCENE(SDFR_RUN)
#undef SDFR_RUN
	default: return hipErrorInvalidValue;
	}
}

template <class Scene, bool DBG>
static hipError_t run_wavefront(const FrameU &U, const RowMap &rm, void *out, int format, uint32_t *pixel_stats, RenderTotals *totals,
	const WavefrontWorkspace &ws, hipStream_t stream, hipEvent_t *march_events, hipEvent_t *shade_events, int *n_rounds_out)
{
	const uint32_t n_work = work_items(U, rm);
	if ((size_t)n_work > ws.capacity) return hipErrorInvalidValue; // lists are indexed by work item, state by pixel id < n_work
	const uint32_t init_blocks = (n_work + SDFR_BLOCK - 1) / SDFR_BLOCK;
	hipLaunchKernelGGL(k_init, dim3(init_blocks), dim3(SDFR_BLOCK), 0, stream, U, rm, n_work, ws, pixel_stats);

	int device = 0;
	(void)hipGetDevice(&device);
	const int cus = device_cu_count(device);
	int march_blocks_per_cu = 0, shade_blocks_per_cu = 0;
	(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&march_blocks_per_cu, k_march<Scene, DBG>, SDFR_BLOCK, 0);
	(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&shade_blocks_per_cu, k_shade<Scene, DBG>, SDFR_BLOCK, 0);
	if (march_blocks_per_cu < 1) march_blocks_per_cu = 1;
	if (shade_blocks_per_cu < 1) shade_blocks_per_cu = 1;
	// persistent grids: what the occupancy query calls resident (it may over-state by a block
	// per CU for SGPR-heavy kernels; harmless here because work is claimed dynamically), but
	// never more waves than there are ranges to claim
	const uint32_t grabs = (n_work + SDFR_GRAB - 1u) / SDFR_GRAB;
	uint32_t march_blocks = (uint32_t)(cus * march_blocks_per_cu);
	if (march_blocks > (grabs + 3u) / 4u) march_blocks = (grabs + 3u) / 4u;
	if (march_blocks < 1) march_blocks = 1;
	uint32_t shade_blocks = (uint32_t)(cus * shade_blocks_per_cu);
	if (shade_blocks > init_blocks) shade_blocks = init_blocks;
	if (shade_blocks < 1) shade_blocks = 1;

	uint32_t *list_cur = ws.list_a, *list_next = ws.list_b;
	const int rounds = U.bounce_count;
	for (int r = 0; r < rounds; ++r)
	{
		if (march_events) (void)hipEventRecord(march_events[2 * r], stream);
		hipLaunchKernelGGL((k_march<Scene, DBG>), dim3(march_blocks), dim3(SDFR_BLOCK), 0, stream, U, ws, list_cur, ws.counters + r,
			ws.counters + 32 + r, pixel_stats, totals);
		if (march_events) (void)hipEventRecord(march_events[2 * r + 1], stream);
		if (shade_events) (void)hipEventRecord(shade_events[2 * r], stream);
		hipLaunchKernelGGL((k_shade<Scene, DBG>), dim3(shade_blocks), dim3(SDFR_BLOCK), 0, stream, U, rm, ws, list_cur, ws.counters + r, list_next,
			ws.counters + r + 1, r, out, format, pixel_stats, totals);
		if (shade_events) (void)hipEventRecord(shade_events[2 * r + 1], stream);
		uint32_t *t = list_cur;
		list_cur = list_next;
		list_next = t;
	}
	if (n_rounds_out) *n_rounds_out = rounds;
	return hipGetLastError();
}

hipError_t launch_wavefront_schedule(int scene, const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream, hipEvent_t *march_events, hipEvent_t *shade_events,
	int *n_rounds_out)
{
	switch (scene)
	{
#define SDFR_RUN(I, S) case I: return frame_needs_debug(U) ? run_wavefront<S, true>(U, rows, out, format, pixel_stats, totals, ws, stream, march_events, shade_events, n_rounds_out) : run_wavefront<S, false>(U, rows, out, format, pixel_stats, totals, ws, stream, march_events, shade_events, n_rounds_out);
		SDFR_FOR_EACH_SCENE(SDFR_RUN)
#undef SDFR_RUN
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_assemble_strips(int width, int height, int world, const void *gathered, void *out_image, int format, int priv_count,
	int priv_period, hipStream_t stream)
{
	const size_t strips = ((size_t)height + 7) / 8 - private_strip_count((uint32_t)(((size_t)height + 7) / 8), priv_count, priv_period);
	const size_t strip_pixels = ((strips + world - 1) / world) * 8 * (size_t)width;
	hipLaunchKernelGGL(k_assemble, dim3((width + SDFR_BLOCK - 1) / SDFR_BLOCK, height), dim3(SDFR_BLOCK), 0, stream, width, height, world, strip_pixels,
		reinterpret_cast<const uint32_t *>(gathered), reinterpret_cast<uint32_t *>(out_image), format, priv_count, priv_period);
	return hipGetLastError();
}

} // namespace sdfr
