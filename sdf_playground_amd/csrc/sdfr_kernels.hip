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

#include <cstdlib>
#include "sdfr_perpixel.h"
#include "sdfr_pixel_kernel.h"

namespace sdfr {

uint32_t launch_work_items(int width, const RowMap &rm)
{
	const uint32_t tw_log2 = (uint32_t)rm.tile_w_log2, th_log2 = 6u - tw_log2;
	const uint32_t tiles_x = ((uint32_t)width + (1u << tw_log2) - 1u) >> tw_log2;
	const uint32_t tiles_y = ((uint32_t)rm.local_rows + (1u << th_log2) - 1u) >> th_log2;
	return tiles_x * tiles_y * 64u;
}
// what a handle's per-launch scratch is sized for: the work items, or -- a persistent launch hands a full frame out in squares of tiles that
// cover it with a margin (row_map_units), and launches at most one block per tile handed out -- the items of those squares
uint32_t launch_capacity_items(int width, const RowMap &rm)
{
	const uint32_t items = launch_work_items(width, rm);
	RowMap units = rm;
	row_map_tiles(units, width);
	row_map_units(units, SDFR_ROW_FEEDBACK_MAX);
	const uint32_t padded = units.unit_log2 ? (units.units << (2u * units.unit_log2)) * 64u : 0u;
	return padded > items ? padded : items;
}

// =================================================================================================
// kernels that do not depend on the scene (the per-scene ones: sdfr_kernels_group.hip)
// =================================================================================================
// Folds the per-block partial sums of a pixel-schedule launch (one 32-byte record per wave: 4 MB
// at 4K) into the render totals, which the pixel kernel has cleared: a few blocks, each keeping
// several independent loads in flight per thread, then ONE set of four atomics per block.
#define SDFR_REDUCE_THREADS 256
#define SDFR_REDUCE_BLOCKS 64
__global__ __launch_bounds__(SDFR_REDUCE_THREADS) void k_reduce_totals(const RenderTotals *__restrict__ partials, uint32_t n, RenderTotals *totals,
	uint32_t *tile_cursors, uint32_t feedback_rows, unsigned long long frame_pixels, uint32_t feedback_key)
{
	// the pixel kernel before this one has drained its tile cursors: back to zero for the next launch
	if (blockIdx.x == 0 && threadIdx.x < SDFR_TILE_CURSORS) tile_cursors[threadIdx.x * SDFR_TILE_CURSOR_STRIDE] = 0u;
	// Row feedback (sdfr_pixel_kernel.h): the last block does nothing but sort the tile rows by this frame's cost, dearest
	// first, into the order the next frame's launch hands them out in.  A counting sort into 64 cost classes (round 3; round
	// 2 ranked every row against every other: 270 x 270 comparisons were most of this kernel's 18 us) -- the order only has
	// to be a permutation that starts with the dear rows.
	if (blockIdx.x == gridDim.x - 1u && gridDim.x > 1u)
	{
		constexpr uint32_t BUCKETS = 64u;
		__shared__ uint32_t cost[SDFR_ROW_FEEDBACK_MAX];
		__shared__ uint32_t bucket_fill[BUCKETS], bucket_base[BUCKETS];
		__shared__ unsigned long long frame_rays;
		__shared__ uint32_t max_cost;
		uint32_t *meta = tile_cursors + SDFR_ROW_META, *row_cost = tile_cursors + SDFR_ROW_COST, *row_order = tile_cursors + SDFR_ROW_ORDER;
		uint32_t *row_rays = tile_cursors + SDFR_ROW_RAYS;
		uint32_t rows = feedback_rows <= SDFR_ROW_FEEDBACK_MAX ? feedback_rows : 0u;
		if (threadIdx.x == 0)
		{
			frame_rays = 0ull;
			max_cost = 0u;
		}
		if (threadIdx.x < BUCKETS) bucket_fill[threadIdx.x] = 0u;
		__syncthreads();
		unsigned long long my_rays = 0ull;
		uint32_t my_max = 0u;
		for (uint32_t i = threadIdx.x; i < SDFR_ROW_FEEDBACK_MAX; i += SDFR_REDUCE_THREADS)
		{
			cost[i] = i < rows ? row_cost[i] : 0u;
			my_rays += i < rows ? row_rays[i] : 0u;
			my_max = cost[i] > my_max ? cost[i] : my_max;
			row_cost[i] = 0u;
			row_rays[i] = 0u;
		}
		atomicAdd(&frame_rays, my_rays);
		atomicMax(&max_cost, my_max);
		__syncthreads();
		// many rays per pixel: leave the rows in image order (SDFR_ROW_FEEDBACK_MAX_RAYS); every tile added 64 to its row's cost
		if (frame_rays > (unsigned long long)SDFR_ROW_FEEDBACK_MAX_RAYS * frame_pixels) rows = 0u;
		const float scale = (float)(BUCKETS - 1u) / (float)(max_cost ? max_cost : 1u);
		uint32_t my_bucket[SDFR_ROW_FEEDBACK_MAX / SDFR_REDUCE_THREADS], my_place[SDFR_ROW_FEEDBACK_MAX / SDFR_REDUCE_THREADS];
#pragma unroll
		for (uint32_t k = 0; k < SDFR_ROW_FEEDBACK_MAX / SDFR_REDUCE_THREADS; ++k)
		{
			const uint32_t i = k * SDFR_REDUCE_THREADS + threadIdx.x;
			const uint32_t b = (BUCKETS - 1u) - (uint32_t)((float)cost[i] * scale); // 0 = the dearest class
			my_bucket[k] = b < BUCKETS ? b : 0u;
			my_place[k] = i < rows ? atomicAdd(&bucket_fill[my_bucket[k]], 1u) : 0u;
		}
		__syncthreads();
		// where each class starts: an exclusive scan over the 64 classes, one per lane of the first wave (a loop on one
		// thread was 64 dependent LDS round trips, 3 of this kernel's 7.6 us)
		static_assert(BUCKETS == 64u, "one cost class per lane");
		if (threadIdx.x < BUCKETS)
		{
			const uint32_t fill = bucket_fill[threadIdx.x];
			uint32_t incl = fill;
#pragma unroll
			for (uint32_t off = 1; off < BUCKETS; off <<= 1)
			{
				const uint32_t below = __shfl_up(incl, off);
				if (threadIdx.x >= off) incl += below;
			}
			bucket_base[threadIdx.x] = incl - fill;
		}
		__syncthreads();
#pragma unroll
		for (uint32_t k = 0; k < SDFR_ROW_FEEDBACK_MAX / SDFR_REDUCE_THREADS; ++k)
		{
			const uint32_t i = k * SDFR_REDUCE_THREADS + threadIdx.x;
			if (i < rows) row_order[bucket_base[my_bucket[k]] + my_place[k]] = i;
		}
		if (threadIdx.x == 0) *meta = rows ? feedback_key : 0u; // 0: no order for the next launch
		return;
	}
	const uint32_t fold_blocks = gridDim.x > 1u ? gridDim.x - 1u : 1u;
	__shared__ unsigned long long acc[4];
	if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
	__syncthreads();
	unsigned long long s[4] = {0ull, 0ull, 0ull, 0ull};
	const ulonglong4 *src = reinterpret_cast<const ulonglong4 *>(partials);
	const uint32_t stride = fold_blocks * SDFR_REDUCE_THREADS;
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
int pixel_tile_cursor_words() { return (int)SDFR_CURSOR_WORDS; }

PixelLaunchMode pixel_launch_mode(int launch_mode, bool scene_default_persistent, int scene_retire_after)
{
	static const int env_persistent = [] { const char *e = getenv("SDFR_PIXEL_PERSISTENT"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
	static const int env_blocks = [] { const char *e = getenv("SDFR_PIXEL_BLOCKS_PER_CU"); return e ? atoi(e) : 0; }();
	PixelLaunchMode m;
	m.persistent = launch_mode == 2 || (launch_mode == 0 && scene_default_persistent);
	if (env_persistent >= 0) m.persistent = env_persistent != 0;
	m.blocks_per_cu = env_blocks;
	static const int env_retire = [] { const char *e = getenv("SDFR_PIXEL_RETIRE_AFTER"); return e ? atoi(e) : -1; }();
	m.retire_after = env_retire >= 0 ? env_retire : scene_retire_after;
	return m;
}

uint32_t pixel_launch_blocks(const PixelLaunchMode &mode, uint32_t tiles, uint32_t resident_blocks)
{
	if (!mode.persistent) return tiles;
	uint32_t blocks = resident_blocks;
	if (mode.retire_after > 0) blocks = resident_blocks / 2u + tiles / (uint32_t)mode.retire_after;
	if (blocks < resident_blocks) blocks = resident_blocks;
	return blocks < tiles ? blocks : tiles;
}

uint32_t pixel_feedback_key(uint32_t scene_key, int width, const RowMap &rm, uint32_t feedback_rows)
{
	static_assert(SDFR_ROW_FEEDBACK_MAX < 1024u, "the unit count rides in the key's low 10 bits");
	if (feedback_rows > SDFR_ROW_FEEDBACK_MAX) return 0u; // no feedback for such a launch (the kernel and the fold agree: fb_rows <= MAX)
	// FNV-1a over what a row order depends on; never 0
	uint32_t h = 2166136261u;
	const uint32_t words[] = {scene_key, (uint32_t)width, (uint32_t)rm.local_rows, (uint32_t)rm.rank, (uint32_t)rm.world, (uint32_t)rm.tile_w_log2,
		(uint32_t)rm.priv_count, (uint32_t)rm.priv_period, (uint32_t)rm.direct, rm.unit_log2};
	for (uint32_t w : words)
		for (int b = 0; b < 4; ++b) h = (h ^ ((w >> (8 * b)) & 0xffu)) * 16777619u;
	h = (h & ~1023u) | feedback_rows;
	return h ? h : 1024u;
}

hipError_t launch_reduce_totals(const RenderTotals *partials, uint32_t n_blocks, RenderTotals *totals, hipStream_t stream, uint32_t *tile_cursors,
	uint32_t feedback_rows, unsigned long long frame_pixels, uint32_t feedback_key)
{
	static const bool feedback = [] { const char *e = getenv("SDFR_TILE_FEEDBACK"); return e ? atoi(e) != 0 : true; }(); // developer knob
	// developer knob: rays per pixel above which tile rows stay in image order (default SDFR_ROW_FEEDBACK_MAX_RAYS)
	static const unsigned long long max_rays = [] { const char *e = getenv("SDFR_TILE_FEEDBACK_MAX_RAYS"); return e && atoi(e) > 0 ? (unsigned long long)atoi(e) : (unsigned long long)SDFR_ROW_FEEDBACK_MAX_RAYS; }();
	if (frame_pixels < (~0ull >> 16)) frame_pixels = frame_pixels * max_rays / SDFR_ROW_FEEDBACK_MAX_RAYS;
	uint32_t blocks = (n_blocks + SDFR_REDUCE_THREADS * 4 - 1) / (SDFR_REDUCE_THREADS * 4);
	if (blocks > SDFR_REDUCE_BLOCKS) blocks = SDFR_REDUCE_BLOCKS;
	if (blocks < 1) blocks = 1;
	// one more block: it sorts the tile rows for the next frame while the others fold the counters
	hipLaunchKernelGGL(k_reduce_totals, dim3(blocks + 1), dim3(SDFR_REDUCE_THREADS), 0, stream, partials, n_blocks, totals, tile_cursors,
		feedback ? feedback_rows : 0u, frame_pixels, feedback_key);
	return hipGetLastError();
}

// march result fields, list counters: see sdfr_kernels_group.hip
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

hipError_t launch_wavefront_init(const FrameU &U, const RowMap &rm, uint32_t n_work, const WavefrontWorkspace &ws, uint32_t *pixel_stats, hipStream_t stream)
{
	hipLaunchKernelGGL(k_init, dim3((n_work + SDFR_BLOCK - 1) / SDFR_BLOCK), dim3(SDFR_BLOCK), 0, stream, U, rm, n_work, ws, pixel_stats);
	return hipGetLastError();
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

// the per-scene launchers live in SDFR_GROUPS translation units (sdfr_kernels_group.hip); scene i is in group i % SDFR_GROUPS
#define SDFR_DECLARE_GROUP(G) \
	hipError_t launch_pixel_group##G(int, const FrameU &, const RowMap &, void *, int, uint32_t *, RenderTotals *, const WavefrontWorkspace &, hipStream_t, int); \
	hipError_t launch_wavefront_group##G(int, const FrameU &, const RowMap &, void *, int, uint32_t *, RenderTotals *, const WavefrontWorkspace &, hipStream_t, \
		hipEvent_t *, hipEvent_t *, int *);
SDFR_FOR_EACH_GROUP(SDFR_DECLARE_GROUP)
#undef SDFR_DECLARE_GROUP

int scene_tile_w_log2(int scene)
{
	switch (scene)
	{
#define SDFR_SHAPE(I, S) case I: return SceneTileShape<S>::value;
		SDFR_FOR_EACH_SCENE(SDFR_SHAPE)
#undef SDFR_SHAPE
	default: return 3;
	}
}

hipError_t launch_pixel_schedule(int scene, const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream, int launch_mode)
{
	if (scene < 0 || scene >= SDFR_SCENE_COUNT) return hipErrorInvalidValue;
	switch (scene % SDFR_GROUPS)
	{
#define SDFR_CALL_GROUP(G) case G: return launch_pixel_group##G(scene, U, rows, out, format, pixel_stats, totals, ws, stream, launch_mode);
		SDFR_FOR_EACH_GROUP(SDFR_CALL_GROUP)
#undef SDFR_CALL_GROUP
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_wavefront_schedule(int scene, const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream, hipEvent_t *march_events, hipEvent_t *shade_events,
	int *n_rounds_out)
{
	if (scene < 0 || scene >= SDFR_SCENE_COUNT) return hipErrorInvalidValue;
	switch (scene % SDFR_GROUPS)
	{
#define SDFR_CALL_GROUP(G) case G: return launch_wavefront_group##G(scene, U, rows, out, format, pixel_stats, totals, ws, stream, march_events, shade_events, n_rounds_out);
		SDFR_FOR_EACH_GROUP(SDFR_CALL_GROUP)
#undef SDFR_CALL_GROUP
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
