// sdfr_render_pixel.h -- the whole pipeline for ONE pixel, start to finish, on one lane.
//
// Used by the "pixel" kernel (one lane per pixel, sdfr_pixel_kernel.h), by scenes compiled at
// run time (sdfr_jit.cpp) and by tests/hostsim, which compiles it for the CPU to bit-compare the
// pipeline stages with the oracle where no GPU is available.  The schedule follows the
// reference's bounce loop (pshader_sdf.hlsl:286-634) literally.
#pragma once
#include "sdfr_pixel.h"

namespace sdfr {


struct LocalRayStore
{
	RayRec slot[SDFR_MAX_RAYS];
	SDF_HD void put(int i, const RayRec &r) { slot[i] = r; }
	SDF_HD RayRec get(int i) const { return slot[i]; }
	SDF_HD void ray_marched(const RayRec &, uint32_t, int) {}
};

// Write-behind cache of depth one in front of a ray store: the most recently pushed ray stays
// in registers and reaches the backing store only when another push follows.  Most pixels
// spawn a single child (the shadow ray) that is popped right away, so its 44-byte record never
// travels to HBM and back.
//
// A store also keeps the pixel's ray footprint (PixelRay) between the prologue and the hit
// shading that needs it for the checker filter: keep_pixel_ray / pixel_ray_kept; and the pixel's
// tone-map flag, which only hit shading touches: keep_hdr / hdr_kept.
template <class Backing>
struct CachedRayStore
{
	Backing &backing;
	RayRec cached;
	int cached_slot;
	PixelRay kept;
	float hdr;
	SDF_HD explicit CachedRayStore(Backing &b) : backing(b), cached_slot(-1) {}
	SDF_HD void keep_pixel_ray(const PixelRay &pr) { kept = pr; }
	SDF_HD PixelRay pixel_ray_kept() const { return kept; }
	SDF_HD void keep_hdr(float h) { hdr = h; }
	SDF_HD float hdr_kept() const { return hdr; }
	SDF_HD void ray_marched(const RayRec &r, uint32_t iter, int status) { backing.ray_marched(r, iter, status); }
	SDF_HD void put(int i, const RayRec &r)
	{
		if (cached_slot >= 0) backing.put(cached_slot, cached);
		cached = r;
		cached_slot = i;
	}
	SDF_HD RayRec get(int i)
	{
		if (i == cached_slot)
		{
			cached_slot = -1;
			return cached;
		}
		return backing.get(i);
	}
};

// Q1 of SURVEY.md 8: the reference takes the forward-difference normal of every hit, also of
// shadow-ray hits, whose shading never looks at it -- unless the scene's material callback reads
// MaterialInput.obj_normal.  A scene that declares `shadow_hits_need_normal = false` lets the
// pixel kernel skip those three scene evaluations; outputs and counters are unchanged.  Scenes
// without the declaration (run-time scenes by default) keep the reference's behaviour.
template <class...>
struct VoidOf { typedef void type; };
template <class Scene, class = void>
struct ShadowHitsNeedNormal { static constexpr bool value = true; };
template <class Scene>
struct ShadowHitsNeedNormal<Scene, typename VoidOf<decltype(Scene::shadow_hits_need_normal)>::type>
{
	static constexpr bool value = Scene::shadow_hits_need_normal;
};

// How a scene's pixel kernel is launched by default (sdfr_set_launch_mode(AUTO)): a scene that declares
// `persistent_tiles = true` gets the persistent launch (resident waves pull tiles, TileQueue), the others one
// wave per tile.  Measured per scene on MI355X at 3840x2160 (profiles/r02_launch_modes.txt): the persistent
// launch wins where tiles are expensive and uneven (labyrinth, cube_sea, fractal, lense, light_shadows: 1.5-3 %)
// and loses where they are cheap (fast_sphere +14 %, cube +33 %: the tile cursors become the bottleneck).
template <class Scene, class = void>
struct PersistentTiles { static constexpr bool value = false; };
template <class Scene>
struct PersistentTiles<Scene, typename VoidOf<decltype(Scene::persistent_tiles)>::type>
{
	static constexpr bool value = Scene::persistent_tiles;
};

// A scene may say `static constexpr bool square_units = true;`: its persistent full-frame launches hand the tiles out in squares (RowMap::
// unit_log2) instead of tile rows, dearest square first by the last frame's cost -- for a scene that is an object in the middle of the picture,
// whose dear tiles sit in the middle of many rows (fractal: BASELINE configuration 4 1.174 -> 1.062 ms; the labyrinth +1.6 %, cube_sea +2.8 %,
// lense with 8 lights +13 ... +17 %: their cost runs along rows; gems with 8 lights 1.657 -> 1.563 ms; the tree gains 2 % from the squares and
// loses 7 % by carrying the code: another register draw).
template <class Scene, class = void>
struct SquareUnits { static constexpr bool value = false; };
template <class Scene>
struct SquareUnits<Scene, typename VoidOf<decltype(Scene::square_units)>::type> { static constexpr bool value = Scene::square_units; };

// The pixels of a wave: 8 x 8 unless the scene says `static constexpr int tile_w_log2 = n;` (a (1 << n) x (64 >> n) tile, n = 3 .. 6).  Measured
// for every scene and configuration (profiles/r03_launch_experiments.txt): 8 x 8 is the best or within 1.5 % -- except gems with 8 lights and depth 4,
// whose waves lose fewer lanes on 16 x 4 (BASELINE configuration 5g 1.579 -> 1.517 ms), and lense and neon at 4K (-3 %).
template <class Scene, class = void>
struct SceneTileShape { static constexpr int value = 3; };
template <class Scene>
struct SceneTileShape<Scene, typename VoidOf<decltype(Scene::tile_w_log2)>::type> { static constexpr int value = Scene::tile_w_log2; };

// tiles a wave of the scene's persistent launch renders before it makes room for a younger one (pixel_launch_blocks,
// sdfr_kernels.h); a scene may say `static constexpr int retire_after = n;` (0 = never)
template <class Scene, class = void>
struct RetireAfter { static constexpr int value = 8; };
template <class Scene>
struct RetireAfter<Scene, typename VoidOf<decltype(Scene::retire_after)>::type> { static constexpr int value = Scene::retire_after; };

struct PixelCounters
{
	uint32_t rays, march_evals, hits;
#ifdef SDFR_PHASE_CLOCKS
	// developer build (tools/phase_clocks.py): wave clock spent marching / taking normals / shading
	uint64_t clk_march, clk_grad, clk_shade, clk_miss, clk_total;
#endif
};
#ifdef SDFR_PHASE_CLOCKS
#define SDFR_CLK(var) const uint64_t var = __builtin_readcyclecounter()
#define SDFR_CLK_ADD(field, t0, t1) cnt.field += (t1) - (t0)
#else
#define SDFR_CLK(var)
#define SDFR_CLK_ADD(field, t0, t1)
#endif

// `store` holds the pixel's pending rays (put/get by slot).  The primary ray never enters
// it: the reference pops it from slot 0 before anything is pushed (pshader_sdf.hlsl:289-294),
// so the queue is empty -- and slot 0 free again -- when the first hit is shaded.
template <class Scene, bool DBG, class Store>
SDF_HD vec4 render_pixel(const FrameU &U, int px, int py, PixelCounters &cnt, Store &store)
{
	SDFR_CLK(c_begin);
	constexpr bool INL = !DBG && InlineEscapedShadows<Scene>::value;
	const DebugFlags F = debug_flags(U);
	RayRec ray;
	{
		const PixelRay pr = pixel_ray(U, px, py);
		ray = primary_ray(U, pr);
		store.keep_pixel_ray(pr);
	}
	uint64_t depths = SDFR_QUEUE_EMPTY;
	int count = 0; // rays waiting in the store

	store.keep_hdr(-1.f); // hdr_output "not set" (pshader_sdf.hlsl:284)
	vec3 acc = V3s(0.f);
	for (int bounce = 0; bounce < U.bounce_count; ++bounce)
	{
		if (bounce > 0)
		{
			if (count == 0) break;
			const int idx = queue_next(depths, U.ray_count);
			ray = store.get(idx);
			depths = queue_set_depth(depths, idx, RAY_DEPTH_INVALID);
			--count;
		}
		cnt.rays++;

		const typename Scene::RayInv R = Scene::ray_setup(U, ray.dir, ray_flags(ray));
		const float inside_sign = ray_inside_sign(ray);
		const float max_range = ray_is_shadow(ray) ? ray.shadow_range : U.range;

		SDFR_CLK(c0);
		March m = march_begin(ray.pos, ray.dir);
		int status;
		const uint32_t evals_before = cnt.march_evals;
		const bool shortcuts = !DBG && (RayEscapes<Scene>::available || EscapesFrom<Scene>::available) && U.step_shortcuts != 0 && inside_sign > 0.f;
		float clear_from = 3e38f;
		if constexpr (EscapesFrom<Scene>::available)
			if (shortcuts) clear_from = EscapesFrom<Scene>::get(U, ray.pos, ray.dir, max_range); // (a ray inside a refracting body ends on its surface)
		do
		{
			march_pre(m);
			// Step shortcut.  Only a sample the march will not take back may end the ray: an unrelaxed one (factor 1: the
			// first three samples and everything after a rewind) is final as it stands, a relaxed one once its own distance
			// shows that it did not over-step (march_advance's test) -- an over-stepped sample can lie beyond an obstacle
			// that the rewound march then hits.
			bool escaped = shortcuts && RayEscapes<Scene>::test(U, R, march_pos(m), ray.dir);
			if constexpr (EscapesFrom<Scene>::available) escaped = escaped || m.t >= clear_from;
			if (escaped && m.factor == 1.f)
			{
				status = MARCH_MISS; // what the remaining steps would come to
				break;
			}
			float d;
			if constexpr (SceneReadsMarchState<Scene>::value)
			{
				const PixelRay pr = store.pixel_ray_kept();
				GeoStep gs;
				gs.camera_distance = m.t;
				gs.right_off = pr.right_ray;
				gs.bottom_off = pr.bottom_ray;
				d = map_geometry_at<Scene, DBG>(U, F, R, march_pos(m), ray.dir, true, gs) * inside_sign;
			}
			else
				d = map_geometry<Scene, DBG>(U, F, R, march_pos(m), ray.dir, true) * inside_sign;
			cnt.march_evals++;
			if (escaped && !((m.last_d + d) < m.last_d * m.factor))
			{
				status = MARCH_MISS;
				break;
			}
			status = march_advance(m, d, max_range, (uint32_t)U.iter_count, U.dist_eps);
		} while (status == MARCH_CONTINUE);
		SDFR_CLK(c1);
		SDFR_CLK_ADD(clk_march, c0, c1);
		store.ray_marched(ray, cnt.march_evals - evals_before, status); // a hook for analysis builds (tests/hostsim); empty in the kernels

		vec3 out;
		if (status == MARCH_HIT)
		{
			cnt.hits++;
			HitInfo hit;
			hit.pos = march_pos(m);
			hit.t = m.t;
			hit.d = m.d;
			hit.iter = m.iter;
			hit.normal = V3s(0.f);
			hit.sample_dist = U.grad_eps;
			if (ShadowHitsNeedNormal<Scene>::value || !ray_is_shadow(ray))
			{
				// map_normal, then the sampled normal unless the scene supplied one (pshader_sdf.hlsl:318-330)
				NormalOut no;
				if (SceneNormal<Scene>::available)
				{
					const PixelRay pr = store.pixel_ray_kept();
					no = scene_normal<Scene>(U, hit.pos, ray.dir, hit.t, pr.right_ray, pr.bottom_ray);
				}
				else
					no = scene_normal<Scene>(U, hit.pos, ray.dir, hit.t, V3s(0.f), V3s(0.f));
				hit.sample_dist = no.sample_dist;
				hit.normal = no.normal;
				if (!no.use_normal)
				{
					const float baseline = m.d * inside_sign;
					float g0, g1, g2;
					if constexpr (SceneReadsMarchState<Scene>::value)
					{
						// grad() works on the hit's GeometryInput: its camera_distance and offsets stay (pshader_sdf.hlsl:164-177)
						const PixelRay pr = store.pixel_ray_kept();
						GeoStep gs;
						gs.camera_distance = hit.t;
						gs.right_off = pr.right_ray;
						gs.bottom_off = pr.bottom_ray;
						g0 = map_geometry_at<Scene, DBG>(U, F, R, grad_sample_pos(hit.pos, 0, no.sample_dist), ray.dir, false, gs) - baseline;
						g1 = map_geometry_at<Scene, DBG>(U, F, R, grad_sample_pos(hit.pos, 1, no.sample_dist), ray.dir, false, gs) - baseline;
						g2 = map_geometry_at<Scene, DBG>(U, F, R, grad_sample_pos(hit.pos, 2, no.sample_dist), ray.dir, false, gs) - baseline;
					}
					else
					{
						g0 = map_geometry<Scene, DBG>(U, F, R, grad_sample_pos(hit.pos, 0, no.sample_dist), ray.dir, false) - baseline;
						g1 = map_geometry<Scene, DBG>(U, F, R, grad_sample_pos(hit.pos, 1, no.sample_dist), ray.dir, false) - baseline;
						g2 = map_geometry<Scene, DBG>(U, F, R, grad_sample_pos(hit.pos, 2, no.sample_dist), ray.dir, false) - baseline;
					}
					hit.normal = normalize(V3(g0, g1, g2));
				}
			}
#ifdef SDFR_PHASE_CLOCKS
			asm volatile("" : "+v"(hit.normal.x), "+v"(hit.normal.y), "+v"(hit.normal.z));
#endif
			SDFR_CLK(c2);
			SDFR_CLK_ADD(clk_grad, c1, c2);

			Spawner<Store> q(store, depths, count, U.ray_count);
			float hdr = store.hdr_kept();
			if constexpr (INL)
			{
				// escaped shadow rays deliver their light from the light loop (InlineEscapedShadows, sdfr_pixel.h): each is a ray and a turn of this loop
				InlineShadows inl;
				inl.acc = acc;
				inl.budget = U.bounce_count - 1 - bounce;
				inl.taken = 0;
				out = shade_hit<Scene, DBG, Store, true>(U, F, ray, store.pixel_ray_kept(), hit, max_range, hdr, q, &inl);
				acc = inl.acc;
				bounce += inl.taken;
				cnt.rays += (uint32_t)inl.taken;
			}
			else
				out = shade_hit<Scene, DBG, Store>(U, F, ray, store.pixel_ray_kept(), hit, max_range, hdr, q);
			store.keep_hdr(hdr);
			depths = q.depths;
			count = q.count;
#ifdef SDFR_PHASE_CLOCKS
			asm volatile("" : "+v"(out.x), "+v"(out.y), "+v"(out.z));
#endif
			SDFR_CLK(c3);
			SDFR_CLK_ADD(clk_shade, c2, c3);
			if constexpr (!INL) acc = acc + out;
		}
		else
		{
			out = shade_miss<Scene>(U, ray, m.iter);
#ifdef SDFR_PHASE_CLOCKS
			asm volatile("" : "+v"(out.x), "+v"(out.y), "+v"(out.z));
#endif
			SDFR_CLK(c4);
			SDFR_CLK_ADD(clk_miss, c1, c4);
			acc = acc + out;
		}
	}
	SDFR_CLK(c_end);
	SDFR_CLK_ADD(clk_total, c_begin, c_end);
	return V4(acc.x, acc.y, acc.z, abs1(store.hdr_kept()));
}

} // namespace sdfr
