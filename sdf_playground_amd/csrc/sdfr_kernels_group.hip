// sdfr_kernels_group.hip -- the per-scene kernels of the renderer (k_pixel, k_march, k_shade; see
// sdfr_kernels.hip for the two schedules), instantiated for ONE group of scenes.  The build compiles
// this file SDFR_GROUPS times with -DSDFR_GROUP=0.. (sdf_playground_amd/buildlib.py), in parallel:
// 23 scenes x 2 debug variants x 3 kernels in one translation unit took a minute to compile, and
// nothing in one scene's kernels depends on another's.  Scene i belongs to group i % SDFR_GROUPS; with SDFR_GROUPS >= the number
// of scenes (sdfr_perpixel.h) every scene is a unit of its own and can be built with its own options (buildlib.SCENE_FLAGS).
#include <cstdlib>

#include "sdfr_kernels.h"
#include "sdfr_perpixel.h"
#include "sdfr_pixel_kernel.h"

#ifndef SDFR_GROUP
#error "compile with -DSDFR_GROUP=<0 .. SDFR_GROUPS-1>"
#endif

namespace sdfr {

// refill a march wave once this many lanes are idle (or when all are)
#define SDFR_REFILL_THRESHOLD 16
// list entries a march wave claims per atomic
#define SDFR_GRAB 128

static uint32_t work_items(const FrameU &U, const RowMap &rm) { return launch_work_items(U.width, rm); }

// =================================================================================================
// PIXEL schedule (body: sdfr_pixel_kernel.h)
// =================================================================================================
template <class Scene, bool DBG>
__global__ SDFR_PIXEL_KERNEL_ATTRS(Scene) void k_pixel(PixelKernelArgs args)
{
	pixel_kernel<Scene, DBG>(args);
}

// =================================================================================================
// WAVEFRONT schedule (k_init lives in sdfr_kernels.hip)
// =================================================================================================
// march result fields
enum { RS_STATUS = 0, RS_T, RS_D, RS_NX, RS_NY, RS_NZ, RS_SAMPLE_DIST, RS_COUNT };
// counters[]: [r] = size of round r's list (r = 0..16); [32 + r] = march cursor of round r
enum { CNT_ROUND0 = 0 };


// ---- k_march ----------------------------------------------------------------------------------------
enum { LANE_IDLE = 0, LANE_MARCH = 1, LANE_GRAD0 = 2, LANE_GRAD1 = 3, LANE_GRAD2 = 4 };

template <class Scene, bool DBG>
__global__ __launch_bounds__(SDFR_BLOCK) void k_march(FrameU U, RowMap rm, WavefrontWorkspace ws, const uint32_t *__restrict__ list,
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
	float sample_dist = U.grad_eps; // of the hit whose normal is being sampled (the scene's map_normal may widen it)
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
			if (state == LANE_GRAD0) p = grad_sample_pos(hp, 0, sample_dist);
			if (state == LANE_GRAD1) p = grad_sample_pos(hp, 1, sample_dist);
			if (state == LANE_GRAD2) p = grad_sample_pos(hp, 2, sample_dist);
			const float dist = map_geometry<Scene, DBG>(U, F, R, p, m.dir, marching);

			if (marching)
			{
				evals++;
				const int status = march_advance(m, dist * inside_sign, max_range, (uint32_t)U.iter_count, U.dist_eps);
				if (status == MARCH_HIT)
				{
					baseline = m.d * inside_sign;
					state = LANE_GRAD0;
					sample_dist = U.grad_eps;
					if (SceneNormal<Scene>::available)
					{
						// map_normal (pshader_sdf.hlsl:318-330): the scene's own normal ends the ray here, a changed spacing feeds the samples
						int px, py;
						pid_to_pixel(U, rm, pid, px, py);
						const PixelRay pr = pixel_ray(U, px, py);
						const NormalOut no = scene_normal<Scene>(U, march_pos(m), m.dir, m.t, pr.right_ray, pr.bottom_ray);
						sample_dist = no.sample_dist;
						if (no.use_normal)
						{
							ws.result[RS_STATUS * cap + pid] = __uint_as_float((uint32_t)MARCH_HIT << 24 | m.iter);
							ws.result[RS_T * cap + pid] = m.t;
							ws.result[RS_D * cap + pid] = m.d;
							ws.result[RS_NX * cap + pid] = no.normal.x;
							ws.result[RS_NY * cap + pid] = no.normal.y;
							ws.result[RS_NZ * cap + pid] = no.normal.z;
							ws.result[RS_SAMPLE_DIST * cap + pid] = sample_dist;
							tot_evals += evals;
							tot_hits += 1;
							if (pixel_stats) pixel_stats[3 * (size_t)pid + 1] += evals;
							state = LANE_IDLE;
						}
					}
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
				ws.result[RS_SAMPLE_DIST * cap + pid] = sample_dist;
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
				hit.sample_dist = ws.result[RS_SAMPLE_DIST * cap + pid];
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

// =================================================================================================
// launchers of this group's scenes
// =================================================================================================
template <class Scene, bool DBG>
static hipError_t run_pixel(const FrameU &U, const RowMap &rm, void *out, int format, uint32_t *pixel_stats, RenderTotals *totals,
	const WavefrontWorkspace &ws, hipStream_t stream, int launch_mode, int scene_index)
{
	const uint32_t n_work = work_items(U, rm);
	if ((size_t)n_work > ws.capacity) return hipErrorInvalidValue;
	// a persistent launch: as many blocks as stay resident, each pulling tiles until none is left
	// (TileQueue, sdfr_pixel_kernel.h).  The occupancy query may over-state by a block per CU for
	// SGPR-heavy kernels (MI355X_MICROARCH.md); a surplus block simply starts when another has ended.
	static int blocks_per_cu = 0; // per instantiation
	if (blocks_per_cu == 0)
	{
		int n = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_pixel<Scene, DBG>, SDFR_PIXEL_BLOCK, 0) != hipSuccess || n < 1) n = 1;
		blocks_per_cu = n;
	}
	int device = 0;
	(void)hipGetDevice(&device);
	const uint32_t tiles_blocks = (n_work + SDFR_PIXEL_BLOCK - 1) / SDFR_PIXEL_BLOCK;
	const PixelLaunchMode mode = pixel_launch_mode(launch_mode, PersistentTiles<Scene>::value, RetireAfter<Scene>::value);
	uint32_t per_cu = (uint32_t)blocks_per_cu;
	if (mode.blocks_per_cu > 0 && (uint32_t)mode.blocks_per_cu < per_cu) per_cu = (uint32_t)mode.blocks_per_cu;
	RowMap rows = rm;
	uint32_t hand_out_items = n_work;
	static const bool squares_off = [] { const char *e = getenv("SDFR_PIXEL_SQUARE_UNITS"); return e && atoi(e) == 0; }(); // developer knob: tile rows for every scene
	if (mode.persistent && SquareUnits<Scene>::value && !squares_off)
	{
		// full frames are handed out in squares of tiles, dearest square first (RowMap::unit_log2); the squares cover the frame with a margin
		row_map_units(rows, SDFR_ROW_FEEDBACK_MAX);
		if (rows.unit_log2) hand_out_items = (rows.units << (2u * rows.unit_log2)) * 64u;
	}
	if ((size_t)hand_out_items > ws.capacity) return hipErrorInvalidValue; // one counter record per block, at most one block per tile handed out
	const uint32_t blocks = pixel_launch_blocks(mode, (hand_out_items + SDFR_PIXEL_BLOCK - 1) / SDFR_PIXEL_BLOCK, (uint32_t)device_cu_count(device) * per_cu);
	rows.retire_after = mode.persistent ? (uint32_t)mode.retire_after : 0u;
	const uint32_t tiles_x = ((uint32_t)U.width + (1u << rm.tile_w_log2) - 1u) >> rm.tile_w_log2;
	const uint32_t feedback_rows = !mode.persistent ? 0u : rows.unit_log2 ? rows.units : tiles_blocks / tiles_x;
	rows.feedback_key = mode.persistent ? pixel_feedback_key((uint32_t)scene_index * 2u + (DBG ? 1u : 0u), U.width, rows, feedback_rows) : 0u;
	PixelKernelArgs args;
	args.U = U;
	args.rm = rows;
	args.n_work = hand_out_items;
	args.format = format;
	args.out = out;
	args.pixel_stats = pixel_stats;
	args.partials = ws.partials;
	args.totals = totals;
	args.ray_queue = ws.ray_queue;
	args.cap = ws.capacity;
	args.tile_cursors = mode.persistent ? ws.tile_cursors : (uint32_t *)nullptr;
	hipLaunchKernelGGL((k_pixel<Scene, DBG>), dim3(blocks), dim3(SDFR_PIXEL_BLOCK), 0, stream, args);
	// (the fold leaves frames with many rays per pixel in image order, SDFR_ROW_FEEDBACK_MAX_RAYS: a rule about tile ROWS -- their queue records
	// are contiguous in image order --, not about squares, which scatter them either way)
	return launch_reduce_totals(ws.partials, blocks, totals, stream, ws.tile_cursors, feedback_rows,
		rows.unit_log2 ? ~0ull >> 8 : (unsigned long long)n_work, rows.feedback_key);
}

template <class Scene, bool DBG>
static hipError_t run_wavefront(const FrameU &U, const RowMap &rm, void *out, int format, uint32_t *pixel_stats, RenderTotals *totals,
	const WavefrontWorkspace &ws, hipStream_t stream, hipEvent_t *march_events, hipEvent_t *shade_events, int *n_rounds_out)
{
	const uint32_t n_work = work_items(U, rm);
	if ((size_t)n_work > ws.capacity) return hipErrorInvalidValue; // lists are indexed by work item, state by pixel id < n_work
	const uint32_t init_blocks = (n_work + SDFR_BLOCK - 1) / SDFR_BLOCK;
	(void)launch_wavefront_init(U, rm, n_work, ws, pixel_stats, stream);

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
		hipLaunchKernelGGL((k_march<Scene, DBG>), dim3(march_blocks), dim3(SDFR_BLOCK), 0, stream, U, rm, ws, list_cur, ws.counters + r,
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

// a scene outside this group is not instantiated here
template <class Scene, bool InGroup>
struct GroupRunner
{
	static hipError_t pixel(const FrameU &, const RowMap &, void *, int, uint32_t *, RenderTotals *, const WavefrontWorkspace &, hipStream_t, int, int) { return hipErrorInvalidValue; }
	static hipError_t wavefront(const FrameU &, const RowMap &, void *, int, uint32_t *, RenderTotals *, const WavefrontWorkspace &, hipStream_t, hipEvent_t *,
		hipEvent_t *, int *)
	{
		return hipErrorInvalidValue;
	}
};
template <class Scene>
struct GroupRunner<Scene, true>
{
	static hipError_t pixel(const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats, RenderTotals *totals,
		const WavefrontWorkspace &ws, hipStream_t stream, int launch_mode, int scene_index)
	{
		return frame_needs_debug(U) ? run_pixel<Scene, true>(U, rows, out, format, pixel_stats, totals, ws, stream, launch_mode, scene_index)
									: run_pixel<Scene, false>(U, rows, out, format, pixel_stats, totals, ws, stream, launch_mode, scene_index);
	}
	static hipError_t wavefront(const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats, RenderTotals *totals,
		const WavefrontWorkspace &ws, hipStream_t stream, hipEvent_t *march_events, hipEvent_t *shade_events, int *n_rounds_out)
	{
		return frame_needs_debug(U) ? run_wavefront<Scene, true>(U, rows, out, format, pixel_stats, totals, ws, stream, march_events, shade_events, n_rounds_out)
									: run_wavefront<Scene, false>(U, rows, out, format, pixel_stats, totals, ws, stream, march_events, shade_events, n_rounds_out);
	}
};

#define SDFR_CAT2(a, b) a##b
#define SDFR_CAT(a, b) SDFR_CAT2(a, b)

hipError_t SDFR_CAT(launch_pixel_group, SDFR_GROUP)(int scene, const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream, int launch_mode)
{
	switch (scene)
	{
#define SDFR_RUN(I, S) case I: return GroupRunner<S, (I) % SDFR_GROUPS == SDFR_GROUP>::pixel(U, rows, out, format, pixel_stats, totals, ws, stream, launch_mode, I);
		SDFR_FOR_EACH_SCENE(SDFR_RUN)
#undef SDFR_RUN
	default: return hipErrorInvalidValue;
	}
}

hipError_t SDFR_CAT(launch_wavefront_group, SDFR_GROUP)(int scene, const FrameU &U, const RowMap &rows, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream, hipEvent_t *march_events, hipEvent_t *shade_events, int *n_rounds_out)
{
	switch (scene)
	{
#define SDFR_RUN(I, S) case I: return GroupRunner<S, (I) % SDFR_GROUPS == SDFR_GROUP>::wavefront(U, rows, out, format, pixel_stats, totals, ws, stream, march_events, shade_events, n_rounds_out);
		SDFR_FOR_EACH_SCENE(SDFR_RUN)
#undef SDFR_RUN
	default: return hipErrorInvalidValue;
	}
}

} // namespace sdfr
