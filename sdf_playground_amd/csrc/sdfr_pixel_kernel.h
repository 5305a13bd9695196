// sdfr_pixel_kernel.h -- device side of the PIXEL schedule: one lane per pixel runs the
// reference's bounce loop start to finish (sdfr_render_pixel.h), waves cover 8x8 pixel tiles,
// pending rays wait in an HBM queue behind a one-entry register cache.
//
// Shared by the kernels compiled ahead of time (sdfr_kernels.hip) and by the translation unit
// sdfr_jit.cpp builds around a scene compiled at run time (hiprtc), so both run the very same
// code.  Device compilation only.
#pragma once
#include "sdfr_render_pixel.h"

namespace sdfr {

#define SDFR_BLOCK 256
// Threads per block of the pixel kernels.  One wave per block: nothing in the kernel needs a
// larger group, and a block holds its registers and LDS until its LAST wave ends, which with
// step counts this uneven leaves slots idle (measured at 4K, 256 -> 64 threads: labyrinth -1.5 %,
// fractal and lense -4 %, the larger fold of the per-block counters included).
#ifndef SDFR_PIXEL_BLOCK
#define SDFR_PIXEL_BLOCK 64
#endif
// the pixel kernel is ONE WAVE PER BLOCK: lane_now() indexes the LDS columns, a block's tile-queue slot and its counter record
// are the wave's (TileQueue::start(blockIdx.x), partials[blockIdx.x]); two waves in a block would share them
static_assert(SDFR_PIXEL_BLOCK == 64, "the pixel kernel is one wave per block");
#define SDFR_INVALID_PIXEL 0xffffffffu
#ifdef SDFR_MAX_WAVES_PER_BLOCK
static_assert(SDFR_BLOCK <= 64 * SDFR_MAX_WAVES_PER_BLOCK && SDFR_PIXEL_BLOCK <= 64 * SDFR_MAX_WAVES_PER_BLOCK, "per-wave LDS of the scenes");
#endif
// Launch attributes of every pixel kernel.  The register allocator is held to 7 waves per SIMD
// (<= 72 VGPRs; left alone it takes ~100-160 and fits 3-4): the VALU of gfx950 issues one
// instruction per wave every ~8 cycles (tools/ubench: 7.5-8 cycles per instruction at 1 wave/SIMD,
// 2.5-3.3 at 8), so issue-bound code wants residency more than registers.  With the ray cache and
// the pixel footprint in LDS (LdsCachedRayStore) and without SLP vectorisation the march loops
// need ~60 registers; at 72 a dozen dwords spill in the shading code (labyrinth: 12), none in a
// march loop (checked in the ISA).  8 waves (64 VGPRs, 40 spilled dwords) are no faster.
// Measured at 4K (labyrinth / fractal / lense, ms): 3 waves 1.77 / 2.4 / 7.9 (allocator's choice),
// 4: 1.61 / 2.2 / 7.9, 5: 1.55 / 2.16 / 7.8, 6: 1.50 / 2.25 / 8.0, 7: 1.43 / 2.03 / 7.2, 8: 1.5 / 2.1 / 7.2.
// With the persistent launch (round 2) the headline scene no longer cares (labyrinth at 4K, ms: 3 waves 1.39, 4: 1.42,
// 5: 1.40, 6: 1.40, 7: 1.40, 8: 1.41); scenes with very long evaluations prefer fewer waves with more registers and
// say so with `waves_per_simd` (tree: 5), scenes whose march loop fits 64 registers can ask for 8.
#ifndef SDFR_PIXEL_WAVES_PER_EU
#define SDFR_PIXEL_WAVES_PER_EU 7
#endif
template <class Scene, class = void>
struct PixelWavesPerSimd { static constexpr int value = SDFR_PIXEL_WAVES_PER_EU; };
#ifndef SDFR_PIXEL_WAVES_FIXED // developer builds that sweep SDFR_PIXEL_WAVES_PER_EU override the scenes' own choice
template <class Scene>
struct PixelWavesPerSimd<Scene, typename VoidOf<decltype(Scene::waves_per_simd)>::type> { static constexpr int value = Scene::waves_per_simd; };
#endif
#define SDFR_PIXEL_KERNEL_ATTRS(Scene) __launch_bounds__(SDFR_PIXEL_BLOCK) __attribute__((amdgpu_waves_per_eu(PixelWavesPerSimd<Scene>::value)))

// This lane's index in its wave, made where it is needed (three instructions).  A block of the pixel kernel is one wave, so
// this is threadIdx.x -- which arrives in v0 and cannot be re-made: kept for the whole life of a persistent wave it is a vector
// register at the march loop or a scratch slot that every tile reloads.  The opaque zero keeps the compiler from hoisting the
// computation back out of the tile loop.
__device__ __forceinline__ uint32_t lane_now()
{
	uint32_t zero = 0;
	asm volatile("" : "+s"(zero));
	return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
}

// ---- pixel mapping ------------------------------------------------------------------------------
// Work item w -> pixel: consecutive groups of 64 items form an 8x8 tile so that a wave sees
// neighbouring pixels (coherent materials, similar step counts).  The tile shape is a launch
// parameter (RowMap::tile_w_log2): 8x8, 16x4, 32x2 or 64x1.
struct PixelCoord
{
	int px, py;   // in the full frame
	uint32_t pid; // index in this launch's (compact) image
};
// which tile row the ty-th row to be handed out is: top to bottom, or -- `order` -- last frame's
// rows sorted by what they cost (row feedback, below)
__device__ __forceinline__ uint32_t handed_out_row(uint32_t ty, const uint32_t *order) { return order ? order[ty] : ty; }
// lane `lane` of the wave that renders tile `tile` (wave-uniform: the split into row and column is scalar work)
// the unit (square of tiles, RowMap::unit_log2) the `slot`-th unit to be handed out is
__device__ __forceinline__ uint32_t handed_out_unit(uint32_t slot, const uint32_t *order) { return order ? order[slot] : slot; }
// UNITS: the kernel of a scene that may be handed out in squares (SquareUnits, sdfr_render_pixel.h); the others do not carry the code
template <bool UNITS = false>
__device__ __forceinline__ bool tile_to_pixel(const FrameU &U, const RowMap &rm, uint32_t tile, uint32_t lane, PixelCoord &pc, const uint32_t *order = nullptr)
{
	const uint32_t tw_log2 = (uint32_t)rm.tile_w_log2, th_log2 = 6u - tw_log2;
	uint32_t tx, ty;
	if (UNITS && rm.unit_log2)
	{
		// hand-out index -> unit slot and tile within the unit (raster order inside the square); tiles of an edge unit may lie outside the frame
		const uint32_t ul = rm.unit_log2, within = tile & ((1u << (2u * ul)) - 1u);
		const uint32_t unit = handed_out_unit(tile >> (2u * ul), order);
		uint32_t uy, ux;
		split_by_magic(unit, rm.units_x, rm.units_x_magic, uy, ux);
		tx = (ux << ul) + (within & ((1u << ul) - 1u));
		ty = (uy << ul) + (within >> ul);
		if (tx >= rm.tiles_x) return false;
	}
	else
	{
		uint32_t row;
		tile_row_and_column(rm, tile, row, tx);
		ty = handed_out_row(row, order);
	}
	const int px = (int)((tx << tw_log2) + (lane & ((1u << tw_log2) - 1u)));
	const int lrow = (int)((ty << th_log2) + (lane >> tw_log2));
	if (px >= U.width || lrow >= rm.local_rows) return false;
	const int py = (int)(strip_local_to_global(rm, (uint32_t)lrow >> 3) * 8u) + (lrow & 7);
	if (py >= U.height) return false;
	pc.px = px;
	pc.py = py;
	pc.pid = (uint32_t)(rm.direct ? py : lrow) * (uint32_t)U.width + (uint32_t)px;
	return true;
}
__device__ __forceinline__ bool work_to_pixel(const FrameU &U, const RowMap &rm, uint32_t w, PixelCoord &pc) { return tile_to_pixel<false>(U, rm, w >> 6, w & 63u, pc); }
__device__ __forceinline__ void pid_to_pixel(const FrameU &U, const RowMap &rm, uint32_t pid, int &px, int &py)
{
	const uint32_t lrow = pid / (uint32_t)U.width;
	px = (int)(pid - lrow * (uint32_t)U.width);
	py = rm.direct ? (int)lrow : (int)(strip_local_to_global(rm, lrow >> 3) * 8u + (lrow & 7u));
}
// n_pixels = pixels of this launch's (compact) image: the packed strip format keeps its flag
// bytes behind the n_pixels rgb triples
__device__ __forceinline__ void store_pixel(void *out, int format, uint32_t pid, vec4 c, uint32_t n_pixels)
{
	if (format == FORMAT_RGBA32F)
	{
		reinterpret_cast<float4 *>(out)[pid] = make_float4(c.x, c.y, c.z, c.w);
	}
	else if (format == FORMAT_STRIP_RGB32F_A8)
	{
		// lossless 13 bytes per pixel for the inter-GPU gather: alpha is the hdr flag, 0 or 1
		float *rgb = reinterpret_cast<float *>(out) + 3 * (size_t)pid;
		rgb[0] = c.x;
		rgb[1] = c.y;
		rgb[2] = c.z;
		reinterpret_cast<unsigned char *>(out)[12 * (size_t)n_pixels + pid] = c.w != 0.f ? 1 : 0;
	}
	else if (format == FORMAT_STRIP_RGB16F_A8)
	{
		// what the reference's R16G16B16A16_FLOAT target holds, in 7 bytes per pixel for the inter-GPU
		// gather: rgb as three halves (round to nearest even), alpha (0 or 1) as one byte
		unsigned short *rgb = reinterpret_cast<unsigned short *>(out) + 3 * (size_t)pid;
		rgb[0] = __builtin_bit_cast(unsigned short, (_Float16)c.x);
		rgb[1] = __builtin_bit_cast(unsigned short, (_Float16)c.y);
		rgb[2] = __builtin_bit_cast(unsigned short, (_Float16)c.z);
		reinterpret_cast<unsigned char *>(out)[6 * (size_t)n_pixels + pid] = c.w != 0.f ? 1 : 0;
	}
	else
	{
		// round-to-nearest-even conversions (v_cvt_f16_f32), two halves per dword
		const uint32_t hx = __builtin_bit_cast(unsigned short, (_Float16)c.x), hy = __builtin_bit_cast(unsigned short, (_Float16)c.y);
		const uint32_t hz = __builtin_bit_cast(unsigned short, (_Float16)c.z), hw = __builtin_bit_cast(unsigned short, (_Float16)c.w);
		uint2 v;
		v.x = hx | (hy << 16);
		v.y = hz | (hw << 16);
		reinterpret_cast<uint2 *>(out)[pid] = v;
	}
}

// Block-wide sums of the render counters.
//  * block_store_totals: one plain 32-byte store per block into partials[blockIdx.x]; a one-block
//    kernel (k_reduce_totals) folds the partials afterwards.  The pixel schedule launches ~32k
//    blocks per 4K frame: letting each of them add to the same four global counters serialises
//    at the memory side (measured on MI355X: +1.2 ms on a 0.26 ms fast_sphere frame, +0.23 ms on
//    the 1.75 ms labyrinth frame), hence no atomics on this path.
//  * block_add_totals: atomics on the totals, for the persistent kernels of the wavefront
//    schedule (a few thousand blocks per launch).
__device__ __forceinline__ void block_reduce_counters(unsigned long long *acc, uint32_t pixels, uint32_t rays, uint32_t evals, uint32_t hits)
{
	if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
	__syncthreads();
	for (int off = 32; off > 0; off >>= 1)
	{
		pixels += __shfl_down(pixels, off);
		rays += __shfl_down(rays, off);
		evals += __shfl_down(evals, off);
		hits += __shfl_down(hits, off);
	}
	if ((threadIdx.x & 63) == 0)
	{
		atomicAdd(&acc[0], (unsigned long long)pixels);
		atomicAdd(&acc[1], (unsigned long long)rays);
		atomicAdd(&acc[2], (unsigned long long)evals);
		atomicAdd(&acc[3], (unsigned long long)hits);
	}
	__syncthreads();
}
__device__ __forceinline__ void block_store_totals(RenderTotals *partials, uint32_t pixels, uint32_t rays, uint32_t evals, uint32_t hits)
{
	__shared__ unsigned long long acc[4];
	block_reduce_counters(acc, pixels, rays, evals, hits);
	if (threadIdx.x < 4) reinterpret_cast<unsigned long long *>(&partials[blockIdx.x])[threadIdx.x] = acc[threadIdx.x];
}
__device__ __forceinline__ void block_add_totals(RenderTotals *totals, uint32_t pixels, uint32_t rays, uint32_t evals, uint32_t hits)
{
	__shared__ unsigned long long acc[4];
	block_reduce_counters(acc, pixels, rays, evals, hits);
	if (threadIdx.x < 4 && acc[threadIdx.x]) atomicAdd(reinterpret_cast<unsigned long long *>(totals) + threadIdx.x, acc[threadIdx.x]);
}

// ray record field order in the SoA arrays
enum { RF_PX = 0, RF_PY, RF_PZ, RF_DX, RF_DY, RF_DZ, RF_CX, RF_CY, RF_CZ, RF_RANGE, RF_BITS, RF_COUNT };
__device__ __forceinline__ RayRec load_ray(const float *base, size_t cap, uint32_t pid)
{
	RayRec r;
	r.pos = V3(base[RF_PX * cap + pid], base[RF_PY * cap + pid], base[RF_PZ * cap + pid]);
	r.dir = V3(base[RF_DX * cap + pid], base[RF_DY * cap + pid], base[RF_DZ * cap + pid]);
	r.contrib = V3(base[RF_CX * cap + pid], base[RF_CY * cap + pid], base[RF_CZ * cap + pid]);
	r.shadow_range = base[RF_RANGE * cap + pid];
	r.bits = __float_as_uint(base[RF_BITS * cap + pid]);
	return r;
}
__device__ __forceinline__ void store_ray(float *base, size_t cap, uint32_t pid, const RayRec &r)
{
	base[RF_PX * cap + pid] = r.pos.x;
	base[RF_PY * cap + pid] = r.pos.y;
	base[RF_PZ * cap + pid] = r.pos.z;
	base[RF_DX * cap + pid] = r.dir.x;
	base[RF_DY * cap + pid] = r.dir.y;
	base[RF_DZ * cap + pid] = r.dir.z;
	base[RF_CX * cap + pid] = r.contrib.x;
	base[RF_CY * cap + pid] = r.contrib.y;
	base[RF_CZ * cap + pid] = r.contrib.z;
	base[RF_RANGE * cap + pid] = r.shadow_range;
	base[RF_BITS * cap + pid] = __uint_as_float(r.bits);
}

// Pending rays of one pixel in HBM, pixel schedule: one 48-byte record per (slot, pixel) -- three
// 16-byte accesses off ONE address.  (The wavefront kernels keep their queue as structure-of-
// arrays, load_ray / store_ray above, because there a wave reads the same field of consecutive
// pixels; here almost every ray stays in the LDS cache, and per-field arrays cost eleven 64-bit
// addresses that the compiler forms in the prologue and spills.)  Both layouts use the same
// workspace buffer, never within one frame.
struct GlobalRayStore
{
	float *queue;
	size_t cap;
	uint32_t pid;
	// (a 32-bit record index -- the workspace holds fewer than 2^32 records, sdfr_api.cpp -- so that no zero-extended pixel index
	// has to live in a register pair)
	__device__ __forceinline__ float4 *record(int slot) const { return reinterpret_cast<float4 *>(queue) + (size_t)((uint32_t)slot * (uint32_t)cap + pid) * 3; }
	__device__ __forceinline__ void put(int slot, const RayRec &r)
	{
		float4 *rec = record(slot);
		rec[0] = make_float4(r.pos.x, r.pos.y, r.pos.z, r.dir.x);
		rec[1] = make_float4(r.dir.y, r.dir.z, r.contrib.x, r.contrib.y);
		rec[2] = make_float4(r.contrib.z, r.shadow_range, __uint_as_float(r.bits), 0.f);
	}
	__device__ __forceinline__ RayRec get(int slot) const
	{
		const float4 *rec = record(slot);
		const float4 a = rec[0], b = rec[1], c = rec[2];
		RayRec r;
		r.pos = V3(a.x, a.y, a.z);
		r.dir = V3(a.w, b.x, b.y);
		r.contrib = V3(b.z, b.w, c.x);
		r.shadow_range = c.y;
		r.bits = __float_as_uint(c.z);
		return r;
	}
};

// when the wave took its current tile (100-MHz clock): how long a tile takes decides how many the next atomic claims
struct TileTimer
{
	unsigned long long t0;
	__device__ __forceinline__ void tile_start() { t0 = __builtin_amdgcn_s_memrealtime(); }
	__device__ __forceinline__ uint32_t ticks_since_start() const { return (uint32_t)(__builtin_amdgcn_s_memrealtime() - t0); }
};

// The pixel kernel's ray store: the write-behind cache of CachedRayStore and the pixel's ray
// footprint, held in LDS instead of registers ([field][thread]: conflict-free).  Both live across
// the whole bounce loop but are touched only around shading; as registers they cost 21 VGPRs at
// the march loop, i.e. residency (see SDFR_PIXEL_WAVES_PER_EU) or scratch spills that reach HBM.
enum { SDFR_LDS_RAY_FIELDS = 11, SDFR_LDS_PIXEL_RAY_FIELDS = 6 }; // of the footprint only the two offset vectors are used again
// pointer that stays in the LDS address space (a plain float* would decay to a 64-bit flat
// pointer: flat loads/stores and a register pair per precomputed field address)
typedef __attribute__((address_space(3))) float lds_float;
struct LdsCachedRayStore
{
	GlobalRayStore &backing;
	lds_float *lds; // this thread's column of the block's [17][SDFR_PIXEL_BLOCK] array
	int cached_slot;
	__device__ __forceinline__ LdsCachedRayStore(GlobalRayStore &b, float *column) : backing(b), lds((lds_float *)column), cached_slot(-1) {}
	__device__ __forceinline__ void write_rec(const RayRec &r)
	{
		lds[0 * SDFR_PIXEL_BLOCK] = r.pos.x; lds[1 * SDFR_PIXEL_BLOCK] = r.pos.y; lds[2 * SDFR_PIXEL_BLOCK] = r.pos.z;
		lds[3 * SDFR_PIXEL_BLOCK] = r.dir.x; lds[4 * SDFR_PIXEL_BLOCK] = r.dir.y; lds[5 * SDFR_PIXEL_BLOCK] = r.dir.z;
		lds[6 * SDFR_PIXEL_BLOCK] = r.contrib.x; lds[7 * SDFR_PIXEL_BLOCK] = r.contrib.y; lds[8 * SDFR_PIXEL_BLOCK] = r.contrib.z;
		lds[9 * SDFR_PIXEL_BLOCK] = r.shadow_range;
		lds[10 * SDFR_PIXEL_BLOCK] = __uint_as_float(r.bits);
	}
	__device__ __forceinline__ RayRec read_rec() const
	{
		RayRec r;
		r.pos = V3(lds[0 * SDFR_PIXEL_BLOCK], lds[1 * SDFR_PIXEL_BLOCK], lds[2 * SDFR_PIXEL_BLOCK]);
		r.dir = V3(lds[3 * SDFR_PIXEL_BLOCK], lds[4 * SDFR_PIXEL_BLOCK], lds[5 * SDFR_PIXEL_BLOCK]);
		r.contrib = V3(lds[6 * SDFR_PIXEL_BLOCK], lds[7 * SDFR_PIXEL_BLOCK], lds[8 * SDFR_PIXEL_BLOCK]);
		r.shadow_range = lds[9 * SDFR_PIXEL_BLOCK];
		r.bits = __float_as_uint(lds[10 * SDFR_PIXEL_BLOCK]);
		return r;
	}
	__device__ __forceinline__ void ray_marched(const RayRec &, uint32_t, int) {}
	__device__ __forceinline__ void put(int i, const RayRec &r)
	{
		if (cached_slot >= 0) backing.put(cached_slot, read_rec());
		write_rec(r);
		cached_slot = i;
	}
	__device__ __forceinline__ RayRec get(int i)
	{
		if (i == cached_slot)
		{
			cached_slot = -1;
			return read_rec();
		}
		return backing.get(i);
	}
	__device__ __forceinline__ void keep_pixel_ray(const PixelRay &pr)
	{
		lds_float *p = lds + SDFR_LDS_RAY_FIELDS * SDFR_PIXEL_BLOCK;
		p[0 * SDFR_PIXEL_BLOCK] = pr.right_ray.x; p[1 * SDFR_PIXEL_BLOCK] = pr.right_ray.y; p[2 * SDFR_PIXEL_BLOCK] = pr.right_ray.z;
		p[3 * SDFR_PIXEL_BLOCK] = pr.bottom_ray.x; p[4 * SDFR_PIXEL_BLOCK] = pr.bottom_ray.y; p[5 * SDFR_PIXEL_BLOCK] = pr.bottom_ray.z;
	}
	// (the tone-map flag in LDS as well -- one more field -- measured slower: labyrinth 4K 1.199 -> 1.240 ms, profiles/r03_launch_experiments.txt)
	float hdr_reg;
	__device__ __forceinline__ void keep_hdr(float h) { hdr_reg = h; }
	__device__ __forceinline__ float hdr_kept() const { return hdr_reg; }
	__device__ __forceinline__ PixelRay pixel_ray_kept() const
	{
		const lds_float *p = lds + SDFR_LDS_RAY_FIELDS * SDFR_PIXEL_BLOCK;
		PixelRay pr;
		pr.dir = V3s(0.f); // the primary direction is not needed after the prologue
		pr.right_ray = V3(p[0 * SDFR_PIXEL_BLOCK], p[1 * SDFR_PIXEL_BLOCK], p[2 * SDFR_PIXEL_BLOCK]);
		pr.bottom_ray = V3(p[3 * SDFR_PIXEL_BLOCK], p[4 * SDFR_PIXEL_BLOCK], p[5 * SDFR_PIXEL_BLOCK]);
		return pr;
	}
};

// ---- tile hand-out of the persistent pixel kernel ---------------------------------------------------
// The launch holds about as many waves as the chip keeps resident; a wave renders 8x8 tiles until none
// is left.  Why not one workgroup per tile: the hardware deals workgroups to the 32 shader engines in
// strict rotation (block n -> XCD n % 8, engine (n / 8) % 4: tools/wave_trace.py shows exactly 1/32 of
// the blocks on each), so an engine that happens to draw long-running tiles holds up the hand-out while
// the others run dry -- 15-25 % of the wave slots stood empty through the middle of a labyrinth frame
// (slot refill gap 9 us mean against 0.8 us while all tiles are alike).  Pulling tiles from a counter
// has no such coupling.
//   * 8 cursors, 128 bytes apart (one word sustains ~88 hand-outs per us; a 4K frame of a cheap scene
//     needs several hundred): tile t belongs to cursor t % 8, a wave pulls from the cursor of its XCD and
//     moves on to the next cursor once one has run out.  Which cursor a wave starts with is a matter of
//     speed only: any wave may pull from any cursor, and a wave that starts late (the occupancy query
//     over-states residency for some kernels) simply finds less left.
//   * one tile per atomic, unless tiles are cheap: a wave whose last tile took less than 10 us doubles its
//     claim (up to SDFR_TILE_BATCH_MAX tiles per atomic) and falls back to one as soon as a tile takes
//     longer.  A 4K frame of fast_sphere hands out 500 tiles per us, more than the cursors sustain one by
//     one; but a wave must never sit on several EXPENSIVE tiles: with claims of 8 the last waves of a
//     labyrinth frame held 8 tiles each and the frame took 1.6 instead of 1.35 ms.
//   * k_reduce_totals, which follows every launch, puts the cursors back to zero.
#define SDFR_TILE_CURSORS 8
#define SDFR_TILE_CURSOR_STRIDE 32 // uint32 words between two cursors
#ifndef SDFR_TILE_BATCH_MAX
#define SDFR_TILE_BATCH_MAX 8
#endif
#define SDFR_TILE_FAST_TICKS 1000u // of the 100-MHz clock: a tile rendered this quickly (10 us) makes the wave claim more at once
#define SDFR_NO_TILE 0xffffffffu
// Row feedback.  The frame ends when the last wave ends, so the tiles handed out last should be the cheap ones: every
// tile adds its march evaluations to its tile row's cost, the fold kernel that follows the launch sorts the rows by cost
// (dearest first) into the order the NEXT frame's launch hands them out in, and clears the costs.  Frames of a sequence
// resemble each other row by row (sky, horizon, floor) even when the camera turns.  State lives behind the tile cursors:
// [META] key of the launch the order was made by (RowMap::feedback_key: scene, frame size, row selection, tile shape; anything
// else, e.g. 0 before the first frame or after a frame with many rays per pixel: hand out top to bottom),
// [COST ...] this frame's cost per row, [ORDER ...] the permutation.  Persistent launches of up to 512 tile rows.
#ifndef SDFR_ROW_FEEDBACK_MAX
#define SDFR_ROW_FEEDBACK_MAX 512u
#endif
#define SDFR_ROW_META (SDFR_TILE_CURSORS * SDFR_TILE_CURSOR_STRIDE)
#define SDFR_ROW_COST (SDFR_ROW_META + 32u)
#define SDFR_ROW_ORDER (SDFR_ROW_COST + SDFR_ROW_FEEDBACK_MAX)
#define SDFR_ROW_RAYS (SDFR_ROW_ORDER + SDFR_ROW_FEEDBACK_MAX)
#define SDFR_CURSOR_WORDS (SDFR_ROW_RAYS + SDFR_ROW_FEEDBACK_MAX)
// Not for frames with many rays per pixel (above SDFR_ROW_FEEDBACK_MAX_RAYS on average: the depth-4 / 8-light
// configurations): their pixels keep queued rays in HBM, gigabytes per frame, and rows rendered out of image order
// scatter that traffic -- measured +2.7 % on lense with 8 lights and +1.6 % on gems, against -3.6 % on the labyrinth,
// -2 % on cube_sea at 1080p and -7 % on the fractal.  [RAYS ...]: this frame's rays per row, for that decision.
#define SDFR_ROW_FEEDBACK_MAX_RAYS 4u
struct TileQueue
{
	uint32_t *cursors;
	uint32_t n_tiles, n_waves; // of the launch
	uint32_t shard, dead;      // cursor this wave pulls from; bit s: cursor s has run out
	uint32_t next_k, end_k;    // claimed and not yet rendered: tiles shard + 8 k, k in [next_k, end_k)
	uint32_t batch;            // tiles the next atomic claims
	uint32_t own_tile;         // one-wave-per-tile launches (cursors == nullptr): this wave's tile
	__device__ __forceinline__ void start(uint32_t wave)
	{
		own_tile = wave < n_tiles ? wave : SDFR_NO_TILE;
		dead = 0;
		next_k = end_k = 0;
		batch = 1;
		shard = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & (SDFR_TILE_CURSORS - 1); // HW_REG_XCC_ID
	}
	__device__ __forceinline__ uint32_t tiles_of(uint32_t s) const { return s < n_tiles ? (n_tiles - s + SDFR_TILE_CURSORS - 1u) / SDFR_TILE_CURSORS : 0u; }
	__device__ __forceinline__ bool holds_claimed_tiles() const { return next_k < end_k; }
	// how long the last tile took decides how many tiles the next atomic claims
	__device__ __forceinline__ void tile_took(uint32_t ticks)
	{
		batch = ticks < SDFR_TILE_FAST_TICKS ? (batch * 2u > SDFR_TILE_BATCH_MAX ? (uint32_t)SDFR_TILE_BATCH_MAX : batch * 2u) : 1u;
	}
	// wave-uniform; at most one atomic by one lane per batch
	__device__ __forceinline__ uint32_t next()
	{
		if (!cursors) // one wave per tile (the launch has as many waves as tiles): wave v renders tile v
		{
			const uint32_t v = dead ? SDFR_NO_TILE : own_tile;
			dead = 1;
			return v;
		}
		if (next_k < end_k) return shard + SDFR_TILE_CURSORS * next_k++;
		while (dead != (1u << SDFR_TILE_CURSORS) - 1u)
		{
			const uint32_t in_shard = tiles_of(shard);
			const uint32_t want = batch;
			uint32_t k = 0;
			if (lane_now() == 0) k = in_shard ? atomicAdd(cursors + shard * SDFR_TILE_CURSOR_STRIDE, want) : 0u;
			k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
			if (k < in_shard)
			{
				next_k = k + 1u;
				end_k = k + want < in_shard ? k + want : in_shard;
				return shard + SDFR_TILE_CURSORS * k;
			}
			dead |= 1u << shard;
			shard = (shard + 1u) & (SDFR_TILE_CURSORS - 1u);
			next_k = end_k = 0;
		}
		return SDFR_NO_TILE;
	}
};

// body of the pixel kernel; the __global__ wrappers are k_pixel (scenes compiled ahead of time,
// sdfr_kernels_group.hip) and the extern "C" kernels sdfr_jit.cpp generates around a run-time scene
template <class Scene, bool DBG>
__device__ __forceinline__ void pixel_kernel(const PixelKernelArgs &args_by_value)
{
	// The argument block is the kernarg segment.  Read through the by-value parameter, the compiler loads every field it
	// will ever need at the kernel's entry and keeps it for the whole persistent tile loop: scalar registers run out (82 of
	// them spilled to vector lanes on the labyrinth kernel), which costs vector registers, which spill to scratch -- 30
	// dwords per lane, written once per wave, 150 MB of HBM traffic per 4K frame with waves that retire.  Read through
	// THIS pointer -- the same memory, constant address space, so still scalar loads, but laundered through an empty asm so
	// that the compiler no longer knows up front that it may be dereferenced -- a field is loaded where its code runs.
	typedef const PixelKernelArgs __attribute__((address_space(4))) *ArgsPtr;
	ArgsPtr args_ptr = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(args_ptr));
	(void)args_by_value;
	const PixelKernelArgs &A = *(const PixelKernelArgs *)args_ptr;
	const FrameU &U = A.U;
	const RowMap &rm = A.rm;
	const uint32_t n_work = A.n_work;
	void *const out = A.out;
	const int format = A.format;
	uint32_t *const pixel_stats = A.pixel_stats;
	RenderTotals *const partials = A.partials, *const totals = A.totals;
	float *const ray_queue = A.ray_queue;
	const size_t cap = A.cap;
	uint32_t *const tile_cursors = A.tile_cursors;
	// the fold kernel that follows adds into the totals: clear them here (kernel boundary = ordering)
	if (blockIdx.x == 0 && threadIdx.x < 4) reinterpret_cast<unsigned long long *>(totals)[threadIdx.x] = 0ull;
	__shared__ float lds_rays[SDFR_LDS_RAY_FIELDS + SDFR_LDS_PIXEL_RAY_FIELDS][SDFR_PIXEL_BLOCK];
#ifdef SDFR_WAVE_TRACE
	const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
	uint32_t trace_tiles = 0, trace_ticks = 0; // tiles rendered; 10-ns ticks from taking a tile to having rendered it, summed
#endif
	const uint32_t waves_per_block = SDFR_PIXEL_BLOCK / 64u;
	TileQueue tiles = {tile_cursors, n_work >> 6, gridDim.x * waves_per_block, 0u, 0u, 0u, 0u, 1u, 0u};
	tiles.start(blockIdx.x);
	TileTimer age = {0ull};
#ifdef SDFR_PHASE_CLOCKS
	PixelCounters clk = {};
#endif
	// the wave's running counters over its tiles: summed over the lanes after every tile (a handful of
	// shuffles per tile) and kept wave-uniform, so that no lane carries them through the bounce loop
	uint32_t w_pixels = 0, w_rays = 0, w_evals = 0, w_hits = 0, tiles_done = 0;
	// row feedback (see SDFR_ROW_META): which rows come first, where this frame's costs go
	constexpr bool UNITS = SquareUnits<Scene>::value;
	const uint32_t fb_tiles_x = UNITS && rm.unit_log2 ? 1u << (2u * rm.unit_log2) : rm.tiles_x; // tiles per feedback unit (square, or tile row)
	uint32_t fb_rows, fb_rest;
	if (UNITS && rm.unit_log2)
		fb_rows = rm.units;
	else
		tile_row_and_column(rm, n_work >> 6, fb_rows, fb_rest);
	const bool fb_on = tile_cursors != nullptr && fb_rows <= SDFR_ROW_FEEDBACK_MAX && rm.feedback_key != 0u;
	// (one word for the whole wave: read as a scalar, or the pointer below becomes a pair of vector registers)
	const uint32_t fb_meta = fb_on ? (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_cursors[SDFR_ROW_META]) : 0u;
	const uint32_t *row_order = fb_on && fb_meta == rm.feedback_key ? tile_cursors + SDFR_ROW_ORDER : nullptr;
	const uint32_t fb_cost_cap = 0xffffffffu / (fb_tiles_x ? fb_tiles_x : 1u); // a row's cost is a 32-bit sum over its tiles: no wrap
#ifdef SDFR_WAVE_TRACE
	unsigned long long trace_mark = __builtin_amdgcn_s_memrealtime();
	uint32_t trace_first_claim = 0, trace_claims = 0;
#endif
	for (uint32_t tile = tiles.next(); tile != SDFR_NO_TILE; tile = tiles.next())
	{
		age.tile_start();
#ifdef SDFR_WAVE_TRACE
		// from the end of the tile before (the kernel's first instruction, for the first tile) to having the next one
		if (trace_claims++ == 0) trace_first_claim = (uint32_t)(age.t0 - trace_mark);
#endif
		PixelCounters pcnt = {};
		uint32_t pix = 0;
		PixelCoord pc;
		const uint32_t lane = lane_now();
		if (tile_to_pixel<UNITS>(U, rm, tile, lane, pc, row_order))
		{
			GlobalRayStore backing = {ray_queue, cap, pc.pid};
			LdsCachedRayStore store(backing, &lds_rays[0][lane]);
			vec4 v = render_pixel<Scene, DBG, LdsCachedRayStore>(U, pc.px, pc.py, pcnt, store);
			store_pixel(out, format, pc.pid, v, (uint32_t)rm.local_rows * (uint32_t)U.width);
			if (pixel_stats)
			{
				pixel_stats[3 * (size_t)pc.pid + 0] = pcnt.rays;
				pixel_stats[3 * (size_t)pc.pid + 1] = pcnt.march_evals;
				pixel_stats[3 * (size_t)pc.pid + 2] = pcnt.hits;
			}
			pix = 1;
#ifdef SDFR_PHASE_CLOCKS
			clk.clk_march += pcnt.clk_march; clk.clk_grad += pcnt.clk_grad; clk.clk_shade += pcnt.clk_shade; clk.clk_miss += pcnt.clk_miss; clk.clk_total += pcnt.clk_total;
#endif
		}
		uint32_t a = pix, b = pcnt.rays, c = pcnt.march_evals, d = pcnt.hits;
		{
			// (__shfl_xor makes the lane index by itself, and the compiler keeps that one for the life of the wave)
			const uint32_t me = lane_now();
#pragma unroll
			for (uint32_t off = 32; off > 0; off >>= 1)
			{
				const int peer = (int)((me ^ off) << 2);
				a += (uint32_t)__builtin_amdgcn_ds_bpermute(peer, (int)a);
				b += (uint32_t)__builtin_amdgcn_ds_bpermute(peer, (int)b);
				c += (uint32_t)__builtin_amdgcn_ds_bpermute(peer, (int)c);
				d += (uint32_t)__builtin_amdgcn_ds_bpermute(peer, (int)d);
			}
		}
		w_pixels += (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
		w_rays += (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
		w_evals += (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
		w_hits += (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
		if (fb_on && lane_now() == 0)
		{
			uint32_t row;
			if (UNITS && rm.unit_log2)
				row = handed_out_unit(tile >> (2u * rm.unit_log2), row_order);
			else
			{
				uint32_t tile_row, tile_column;
				tile_row_and_column(rm, tile, tile_row, tile_column);
				row = handed_out_row(tile_row, row_order);
			}
			atomicAdd(tile_cursors + SDFR_ROW_COST + row, c + 64u < fb_cost_cap ? c + 64u : fb_cost_cap);
			atomicAdd(tile_cursors + SDFR_ROW_RAYS + row, b);
		}
#ifdef SDFR_WAVE_TRACE
		trace_ticks += age.ticks_since_start();
		trace_mark = __builtin_amdgcn_s_memrealtime();
#endif
		tiles.tile_took(age.ticks_since_start());
		// make room for a younger wave (pixel_launch_blocks, sdfr_kernels.h) -- but never with claimed tiles in hand
		if (rm.retire_after && ++tiles_done >= rm.retire_after && !tiles.holds_claimed_tiles()) break;
#ifdef SDFR_WAVE_TRACE
		trace_tiles++;
#endif
	}
#ifdef SDFR_WAVE_TRACE
	// developer build (tools/wave_trace.py): the per-block record carries when and where the wave ran
	// instead of its counters: {start, end} in 10-ns ticks, HW_ID | XCC_ID << 32, tiles (8 bits) | ticks until it had its first tile (24 bits) << 8 | ticks spent rendering << 32
	// (a wave's life minus those ticks is what it waited for tiles -- the cursor's round trip, the row-order look-up -- and its start-up)
	{
		const uint32_t ev = trace_ticks;
		if (threadIdx.x == 0)
		{
			unsigned long long *rec = reinterpret_cast<unsigned long long *>(&partials[blockIdx.x]);
			rec[0] = trace_t0;
			rec[1] = __builtin_amdgcn_s_memrealtime();
			rec[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
			rec[3] = (unsigned long long)(trace_tiles & 0xffu) | ((unsigned long long)(trace_first_claim & 0xffffffu) << 8) | ((unsigned long long)ev << 32);
		}
		return;
	}
#endif
#ifdef SDFR_PHASE_CLOCKS
	const PixelCounters &c = clk;
	// the totals carry wave clocks instead of counts: pixels <- whole pixel loop, rays <- march,
	// march_evals <- shading of escaped rays (background), hits <- normals + shading of hits; per wave the lane that stayed longest speaks
	{
		uint64_t best = c.clk_total;
		for (int off = 32; off > 0; off >>= 1) { uint64_t o = __shfl_xor(best, off); best = o > best ? o : best; }
		const uint64_t first = __ballot(c.clk_total == best);
		const bool speaker = (threadIdx.x & 63) == (uint32_t)__builtin_ctzll(first);
		block_store_totals(partials, speaker ? (uint32_t)(c.clk_total >> 4) : 0u, speaker ? (uint32_t)(c.clk_march >> 4) : 0u,
			speaker ? (uint32_t)(c.clk_miss >> 4) : 0u, speaker ? (uint32_t)((c.clk_grad + c.clk_shade) >> 4) : 0u);
		return;
	}
#endif
	// the wave's sums are uniform and the block is this wave: lanes 0-3 write one counter each
	{
		const uint32_t l = lane_now();
		const uint32_t v = l == 0 ? w_pixels : l == 1 ? w_rays : l == 2 ? w_evals : w_hits;
		if (l < 4) reinterpret_cast<unsigned long long *>(&partials[blockIdx.x])[l] = (unsigned long long)v;
	}
}

} // namespace sdfr
