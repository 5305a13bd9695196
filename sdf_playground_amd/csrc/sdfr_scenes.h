// sdfr_scenes.h -- ahead-of-time scene functors (device code, host-compilable).
//
// The reference compiles a scene INTO its pixel shader by textual substitution of
// "sdf_scene.hlsl" (pshader_sdf.hlsl:84, Application.cpp:229,320); here each scene is a
// functor instantiated into the kernel templates.  The scene plugin ABI of the reference --
// map / map_normal / map_light / map_background (README.md:114-119) -- maps to:
//   prepare()     host, once per frame: frame-uniform constants (sin/cos of stime, ...)
//   ray_setup()   once per ray: everything in map() that depends on the ray only
//   dist()        map(..., geometry_step = true): scene distance at a point
//   material()    map(..., geometry_step = false): material of the surface at a hit point
//   normal()      map_normal, optional (SceneNormal, sdfr_pixel.h); every scene of the reference leaves it empty, so
//                 none of the scenes here has one -- the diagnostic scene normal_test does (sdfr_scene_debug.h)
//   light()       map_light, one light slot at a time
//   background()  map_background
//
// Scenes: fast_sphere, cube_sea, labyrinth, fractal, lense, gems, light_shadows
// (Engine/shader/scenes/sdf_scene_<name>.hlsl).
#pragma once
#include "sdfr_frame.h"
#include "sdfr_lib.h"

namespace sdfr {

// =========================================================================================
struct SceneFastSphere
{
	static const char *name() { return "fast_sphere"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return ""; }
	static SDF_HD void prepare(FrameU &) {}
	struct RayInv { GroundInv ground; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		return r;
	}
	// floor + one sphere of radius 0.5 about (0, 1, 0)
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 1.51f, V3(0.f, 1.f, 0.f), 0.52f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3 dir, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		return min1(d, sd_sphere_fast(p - V3(0.f, 1.f, 0.f), dir, fast, 0.5f, U.dist_eps));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		if (on_surface(U, sd_sphere(sp.pos - V3(0.f, 1.f, 0.f), 0.5f)))
		{
			m.diffuse = V4(0.2f, 0.7f, 0.2f, 1.f);
			set_rgb(m.specular, 0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneCubeSea
{
	static const char *name() { return "cube_sea"; }
	static constexpr bool persistent_tiles = true; // expensive, uneven tiles: resident waves pulling tiles win (sdfr_render_pixel.h)
	static constexpr int retire_after = 4; // at 4K, 256 steps: 2.98 (2) / 2.88 (4) / 2.98 (8) ms; configuration 2 (1080p) is indifferent, 0.75 ms from never to 8
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return ""; }
	static SDF_HD void prepare(FrameU &) {}
	// The cell-wall guard divides by the ray direction's x and z at every step: per ray, their
	// reciprocals are formed once (IEEE) and the steps use div_c, which is the correctly rounded
	// quotient for numerators 0 or 2^-60 <= |a| <= 2^40 (the wall offsets are multiples of 2^-24
	// below 2.1) and, as far as 250 000 random divisors in [2^-66, 2] times every numerator can
	// tell, any divisor (tools/divsweep_long.py; the ground plane relies on the same fact).
	// Directions closer to an axis than 2^-60, and exact zeros, keep the IEEE division.
	struct RayInv { GroundInv ground; vec2 barrier; vec2 rdir; bool exact_x, exact_z, rising; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		// cell-wall guard (sdf_primitives.hlsl:118-124): which wall the ray runs towards
		r.barrier = (V2(step1(0.f, dir.x), step1(0.f, dir.z)) - 0.5f) * V2(2.01f, 2.01f);
		r.exact_x = abs1(dir.x) >= 8.6736174e-19f; // 2^-60
		r.exact_z = abs1(dir.z) >= 8.6736174e-19f;
		r.rdir = V2(r.exact_x ? rcp1(dir.x) : 0.f, r.exact_z ? rcp1(dir.z) : 0.f);
		r.rising = dir.y >= 0.f;
		return r;
	}
	// Above the cubes' slab (cube_lower_bound: they end at y = 3.65) a ray that does not descend has the floor behind it
	// and every cube below it for good: only the cell guard still stops it, once per cell wall, out to the range of 100 --
	// two thirds of this scene's march steps (sky-bound primary and reflection rays, shadow rays towards the sun once
	// they have cleared the cubes).
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &R, vec3 p, vec3) { return R.rising && p.y > 3.66f; }
	static SDF_HD float guard_quotient(float num, float den, float rden, bool exact)
	{
		if (exact) return div_c(num, den, rden);
		return num / den;
	}
	struct Cell { vec3 cell_pos; float cube; bool is_other; };
	static SDF_HD vec3 cell_position(vec3 p)
	{
		vec2 rep = op_rep_inf(V2(p.x, p.z), V2(2.f, 2.f)); // x / 2 is an exact multiply already
		return V3(rep.x, p.y, rep.y);
	}
	static SDF_HD Cell eval_cell(const FrameU &U, vec3 p)
	{
		Cell c;
		c.cell_pos = cell_position(p);
		const vec2 rep = V2(c.cell_pos.x, c.cell_pos.z);
		vec2 cell_index = (V2(p.x, p.z) - rep) / 2.f;
		vec2 q = cell_index * 0.5f + 0.25f;
		vec2 sometimes = V2(rne1(frac1(q.x)), rne1(frac1(q.y)));
		c.is_other = sometimes.x < 0.5f && sometimes.y < 0.5f;
		float phase = cell_index.x + cell_index.y * 0.3f + U.stime;
		vec2 sc = sincos1(phase);
		// the cube's rotation angle is cos(phase) * 0.4, at most 0.4 in magnitude: the reduction-free sincos
		vec2 rsc = sincos1_small(sc.y * 0.4f);
		vec2 r = rot2(rep, rsc.x, rsc.y);
		vec3 cube_pos = V3(r.x, p.y, r.y);
		float hs = c.is_other ? 0.25f : 0.5f;
		c.cube = sd_box(cube_pos - V3(0.f, 2.f + sc.x, 0.f), V3s(hs)) - 0.15f;
		return c;
	}
	// Every cube lies in the slab 0.35 <= y <= 3.65: its centre bobs at y = 2 + sin(phase), the box has a
	// half height of at most 0.5 and is rounded by 0.15, and the rotation is about the vertical.  sd_box is
	// at least its distance along one axis, so  cube >= max(p.y - 3.65, 0.35 - p.y).  The scene distance
	// is a min() over floor, cube and cell guard: where that bound (with 0.01 of slack for rounding; the
	// quantities are O(100) at most below y = 1024, above it the cube is evaluated as ever) is not below
	// the smaller of floor and guard, the cube -- two sincos, a rotation and a box: 60 % of an evaluation
	// -- cannot lower the minimum and is left out.  That is every step of a ray that travels above the
	// cubes (sky-bound primary and reflection rays, which need the most steps: the guard stops them at
	// every cell wall up to the range of 100) and the first steps of a shadow ray leaving the floor.
	// Checked numerically in tests/test_scene_bounds_cpu.py.
	static SDF_HD float cube_lower_bound(vec3 p, bool *valid)
	{
		*valid = p.y < 1024.f;
		return max1(p.y - 3.66f, 0.34f - p.y);
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3 dir, bool fast)
	{
		const vec3 cell_pos = cell_position(p);
		const vec2 num = R.barrier - V2(cell_pos.x, cell_pos.z);
		const float tx = guard_quotient(num.x, dir.x, R.rdir.x, R.exact_x);
		const float tz = guard_quotient(num.y, dir.z, R.rdir.y, R.exact_z);
		// min() is taken over the same three values as map() does (floor, cube, guard): any order gives the same bits
		const float d = min1(min1(3e38f, ground_dist(p, fast, R.ground)), min1(tx, tz));
		bool valid;
		const float lb = cube_lower_bound(p, &valid);
		if (valid && lb >= d) return d;
		return min1(d, eval_cell(U, p).cube);
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		Cell c = eval_cell(U, sp.pos);
		if (on_surface(U, c.cube))
		{
			if (c.is_other)
			{
				m.diffuse = V4(0.8f, 0.2f, 0.2f, 1.f);
				set_rgb(m.specular, 1.f);
			}
			else
			{
				m.diffuse = V4(0.6f, 0.5f, 0.2f, 1.f);
				set_rgb(m.specular, 1.f);
				m.reflection = V3s(0.25f);
			}
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneLabyrinth
{
	static const char *name() { return "labyrinth"; }
	static constexpr int waves_per_simd = 7; // configuration 3, round 3 (argument block read on demand: 16 spilled registers instead of 30), one session: 1.247 (5) / 1.234 (6) / 1.226 (7) / 1.254 (8) ms; round 2 had 6 ahead of 7 by 3 % (profiles/r03_launch_experiments.txt)
	static constexpr bool persistent_tiles = true; // expensive, uneven tiles: resident waves pulling tiles win (sdfr_render_pixel.h)
	static constexpr int retire_after = 4; // configuration 3, two sessions: 1.252 (1) / 1.236 (2) / 1.224 (3) / 1.224 (4) / 1.233 (5) / 1.237 (8) / 1.243 (16) ms
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return ""; }
	enum { SU_FIRE_SCROLL = 0 };
	static SDF_HD void prepare(FrameU &U) { U.su[SU_FIRE_SCROLL] = U.stime * 3.f; }

	static SDF_HD float fire_cone(vec3 p) { return sd_round_cone(p, V3(0.f, 1.1f, 0.f), V3(0.f, 1.6f, 0.f), 0.15f, 0.1f); }

	struct RayInv { GroundInv ground; bool skip_fire, rising; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &f)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		// a ray continuing through a transparent surface ignores the fire it just left
		// (OBJECT_TRANSPARENT, pshader_sdf.hlsl:80; sdf_scene_labyrinth.hlsl:55,62)
		r.skip_fire = f.has_transparent && fire_cone(f.last_transparent_pos) < U.dist_eps;
		r.rising = dir.y >= 0.f;
		return r;
	}
	// Nothing of the labyrinth reaches above y = 4: the walls are boxes from 0 to 4, the vase ends at 2.4, stick and flame
	// lie in the ball about (5.2, 2.9, 3) of radius 0.9 (see dist).  A ray above that height that does not descend has the
	// floor behind it as well: the sky-bound primary rays of the upper part of the picture (they start above the walls:
	// no step at all) and every shadow ray once it has cleared the walls.
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &R, vec3 p, vec3) { return R.rising && p.y > 4.01f; }

	static SDF_HD float vase(vec3 p)
	{
		float bowl = sd_sphere(p - V3(0.f, 1.89f, 0.f), 0.5f);
		float stem = sd_capped_cylinder(p - V3(0.f, 0.8f, 0.f), 0.75f, 0.2f);
		float foot = sd_box(p - V3(0.f, 0.075f, 0.f), V3(0.4f, 0.075f, 0.4f));
		float cut_top = sd_plane(p - V3(0.f, 1.9f, 0.f), V3(0.f, 1.f, 0.f));
		float cut_low = sd_plane(p - V3(0.f, 1.5f, 0.f), V3(0.f, -1.f, 0.f));
		float d = max1(bowl, cut_low);
		d = op_pipe_merge_c(d, stem, 0.1f, 4.f);
		d = op_pipe_merge_c(d, foot, 0.1f, 4.f);
		d = max1(d, cut_top);
		d = max1(d, -bowl - 0.06f);
		return d;
	}

	// 20 x 20 cells, mirrored into one octant
	static SDF_HD vec3 fold(vec3 p)
	{
		vec2 rep = op_rep_inf_c(V2(p.x, p.z), 20.f, 1.0f / 20.f);
		float wx = abs1(rep.x), wz = abs1(rep.y);
		if (wz > wx) { float t = wx; wx = wz; wz = t; }
		return V3(wx, p.y, wz);
	}
	static SDF_HD float walls(vec3 wp)
	{
		float wall1 = sd_box(wp - V3(3.5f, 2.f, 3.f), V3(1.5f, 2.f, 1.f));
		float wall2 = sd_box(wp - V3(7.f, 2.f, 5.f), V3(3.f, 2.f, 1.f));
		return min1(wall1, wall2);
	}
	// position in the vase's frame (the vase is mirrored about x = 8)
	static SDF_HD vec3 vase_local(vec3 wp) { return V3(abs1(wp.x - 8.f), wp.y, wp.z) - V3(1.f, 0.f, 3.f); }
	// torch: a tilted wooden stick with a flame cone
	struct Torch { float wood, fire; vec3 torch_pos; };
	static SDF_HD Torch torch(vec3 wp)
	{
		Torch t;
		const float torch_angle = 15.f * SDFR_PI / 180.f;
		const float tc = cos1(torch_angle), ts = sin1(torch_angle);
		vec3 tp = wp - V3(5.f, 2.f, 3.f);
		vec3 sp = V3(tp.x * tc - tp.y * ts, tp.x * ts + tp.y * tc, tp.z);
		tp.x = tp.x - 0.3f;
		t.wood = sd_box(sp - V3(0.f, 0.6f, 0.f), V3(0.05f, 0.5f, 0.05f));
		t.fire = fire_cone(tp);
		t.torch_pos = tp;
		return t;
	}

	// Bounding-ball culling of the two small, expensive objects.  The scene distance is a
	// min() over objects, so an object whose distance is provably >= the running minimum can
	// be left out without changing a single bit.  Bounds (checked numerically by
	// tests/test_scene_bounds_cpu.py):
	//   vase: every part lies in the ball (0, 1.2, 0), r 1.35 of its frame; sphere, capped
	//         cylinder and box are exact SDFs, so each is >= |q - c| - 1.35; the two pipe merges
	//         lower the result by at most 0.075 each (op_pipe(a, b) >= min(a, b) - 0.075 for
	//         min(a, b) >= 0.05 with size 0.1, count 4); the final max() only raise it;
	//   torch: stick and flame (exact SDFs) lie in the ball (0.2, 0.9, 0), r 0.9 of the torch frame.
	// 0.01 of slack covers fp32 rounding (the quantities are O(1)).
	static SDF_HD bool beyond(vec3 v, float reach, float radius)
	{
		const float k = reach + radius;
		return dot(v, v) >= k * k;
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		const vec3 wp = fold(p);
		d = min1(d, walls(wp));
		// One test before the two: both bounding balls (vase: centres (7 | 9, 1.2, 3), r 1.51; torch: (5.2, 2.9, 3), r 0.91) lie in
		// the box x >= 4.29, y <= 3.81, |z - 3| <= 1.51 of the folded cell; a point whose distance to that box along one axis is
		// not below the running minimum is at least radius + minimum from both centres, i.e. both tests below would skip
		// anyway (0.01 more of slack here, so the implication survives rounding).  Four instructions instead of eighteen on
		// most steps of most rays.
		if (max1(max1(4.28f - wp.x, wp.y - 3.82f), abs1(wp.z - 3.f) - 1.52f) >= max1(d, 0.f)) return d;
		const vec3 q = vase_local(wp);
		if (!beyond(q - V3(0.f, 1.2f, 0.f), max1(d, 0.f), 1.35f + 0.15f + 0.01f))
			d = min1(d, vase(q));
		if (!beyond(wp - V3(5.f, 2.f, 3.f) - V3(0.2f, 0.9f, 0.f), max1(d, 0.f), 0.9f + 0.01f))
		{
			const Torch t = torch(wp);
			d = min1(d, t.wood);
			d = R.skip_fire ? d : min1(d, t.fire);
		}
		return d;
	}
	// The first object whose surface the point lies on names the material (MATERIAL macro chain,
	// sdf_scene_labyrinth.hlsl:64-98).  Objects are evaluated only as far down the chain as needed,
	// and the vase / torch only where their bounding balls (see dist) allow |distance| < 1e-4.
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		const vec3 wp = fold(sp.pos);
		if (on_surface(U, walls(wp)))
		{
			m.mpos = sp.pos;
			m.id = MAT_MARBLE_LIGHT;
			return;
		}
		const vec3 q = vase_local(wp);
		if (!beyond(q - V3(0.f, 1.2f, 0.f), U.dist_eps, 1.35f + 0.15f + 0.01f) && on_surface(U, vase(q)))
		{
			m.mpos = sp.pos * 3.f;
			m.id = MAT_MARBLE_DARK;
			return;
		}
		if (beyond(wp - V3(5.f, 2.f, 3.f) - V3(0.2f, 0.9f, 0.f), U.dist_eps, 0.9f + 0.01f)) return;
		const Torch t = torch(wp);
		if (on_surface(U, t.wood))
		{
			m.mpos = V3(sp.pos.x, sp.pos.z, sp.pos.y) * 2.f;
			m.id = MAT_WOOD;
		}
		else if (on_surface(U, t.fire))
		{
			m.mpos = t.torch_pos * 3.f - V3(0.f, U.su[SU_FIRE_SCROLL], 0.f);
			m.id = MAT_FIRE;
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneFractal
{
	static const char *name() { return "fractal"; }
	static constexpr bool square_units = true; // an object in the middle of the picture: its dear tiles are handed out first (sdfr_render_pixel.h)
	static constexpr int retire_after = 1; // its tiles are very uneven: configuration 4 with tile rows 1.203 (1) / 1.178 (2) / 1.199 (4) / 1.38 (8) ms, with the squares 1.044 (1) / 1.058 (2) / 1.070 (3) / 1.072 (4) / 1.19 (8)
	static constexpr int waves_per_simd = 7; // configuration 4, round 3, one session: 1.26 (5) / 1.22 (6) / 1.196 (7) / 1.193 (8) ms
	static constexpr bool persistent_tiles = true; // expensive, uneven tiles: resident waves pulling tiles win (sdfr_render_pixel.h)
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return ""; }
	static SDF_HD void prepare(FrameU &) {}
	struct RayInv { GroundInv ground; bool rising; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		r.rising = dir.y >= 0.f;
		return r;
	}
	// every box lies in the unit ball about (0, 1, 0) (see dist): nothing above y = 2, and the floor is behind a ray that does not descend
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 2.01f, V3(0.f, 1.f, 0.f), 1.02f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	// 8-level recursive fold; returns the distance, and the level that first touched
	static SDF_HD float fold(vec3 p, float *level_hit)
	{
		const float size = 1.f;
		vec3 q = p - V3(0.f, 1.f, 0.f);
		float d = 1e30f;
		float scale = 1.f;
		float hit_level = 0.f;
#pragma unroll
		for (int i = 0; i < 8; ++i)
		{
			// scale = 3^i is a compile-time constant after unrolling: div_c is exact for all of them
			// (tests/test_gpu_math.py, SCENE_DIVISORS)
			float nd = div_c(sd_box(q, V3s(size * 0.5f)), scale, 1.0f / scale);
			if (nd < 0.0001f && d > 0.0001f) hit_level = (float)i;
			d = min1(d, nd);
			q = abs(q);
			sort3_desc(q.y, q.x, q.z);
			q.y = q.y - size * 2.f / 3.f;
			q.y = q.y + size / 3.f;
			sort3_desc(q.y, q.x, q.z);
			q.y = q.y - size / 3.f;
			q = q * 3.f;
			scale = scale * 3.f;
		}
		*level_hit = hit_level;
		return d;
	}
	// Every box of the fractal lies in the ball of radius 1 about (0, 1, 0): a box of level i sits
	// at most sum_{k<i} (2/3) 3^-k from the centre and has a half diagonal of (sqrt(3)/2) 3^-i, and
	// 0.866, 0.955, 0.985, ... < 1.  fold() is the exact distance to the union of the boxes (the
	// folds are isometries), so it is >= |p - c| - 1 and can be skipped when that is not below the
	// floor's distance (0.01 of slack; checked numerically in tests/test_scene_bounds_cpu.py).
	// The same bound level by level: levels i .. 7 as a function of the folded, scaled point q_i are the fractal itself
	// with 8 - i levels, shrunk by 3^i -- so they are >= (|q_i| - 1) / 3^i, and once that is not below the running minimum
	// (the floor included) the remaining levels cannot lower it: min() over the same values, cut short.  0.01 of slack in
	// q_i's units (rounding moves q_i by ~1e-6 |q_i|).  Only for the distance: material() wants the level that touched.
	static SDF_HD float fold_below(vec3 p, float d)
	{
		const float size = 1.f;
		vec3 q = p - V3(0.f, 1.f, 0.f);
		float scale = 1.f;
#pragma unroll
		for (int i = 0; i < 8; ++i)
		{
			const float k = max1(d, 0.f) * scale + (1.f + 0.01f);
			if (dot(q, q) >= k * k) return d;
			d = min1(d, div_c(sd_box(q, V3s(size * 0.5f)), scale, 1.0f / scale));
			q = abs(q);
			sort3_desc(q.y, q.x, q.z);
			q.y = q.y - size * 2.f / 3.f;
			q.y = q.y + size / 3.f;
			sort3_desc(q.y, q.x, q.z);
			q.y = q.y - size / 3.f;
			q = q * 3.f;
			scale = scale * 3.f;
		}
		return d;
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		return fold_below(p, min1(3e38f, ground_dist(p, fast, R.ground)));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		float lvl;
		float d = fold(sp.pos, &lvl);
		if (on_surface(U, d))
		{
			m.diffuse.x = 0.9f;
			m.diffuse.y = 0.7f;
			m.diffuse.z = lvl * 0.125f;
			set_rgb(m.specular, 0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneLense
{
	static const char *name() { return "lense"; }
	static constexpr int tile_w_log2 = 4; // 16 x 4 pixels per wave: 4.10 -> 3.98 ms at 4K, configuration 5 unchanged (sdfr_render_pixel.h)
	static constexpr int waves_per_simd = 8; // the march loop fits 64 registers: BASELINE configuration 5 11.5 -> 11.05 ms (sdfr_pixel_kernel.h)
	static constexpr bool persistent_tiles = true; // expensive, uneven tiles: resident waves pulling tiles win (sdfr_render_pixel.h)
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	// the scene's variable tags in source order; slot k of FrameU::scene_var is the k-th distinct name
	static const char *variables()
	{
		return "VAR_xpos(min = -4, max = 4, step = 0.1) VAR_ypos(min = -4, max = 4, step = 0.1) "
			   "VAR_zpos(min = 0, max = 25, step = 0.1) VAR_mixing(min = 0, max = 1, step = 0.05)";
	}
	enum { SV_XPOS = 0, SV_YPOS = 1, SV_ZPOS = 2, SV_MIXING = 3 };
	enum { SU_MIRROR_S = 0, SU_MIRROR_C = 1 };
	static SDF_HD void prepare(FrameU &U)
	{
		vec2 sc = sincos1(U.stime * 0.3f);
		U.su[SU_MIRROR_S] = sc.x;
		U.su[SU_MIRROR_C] = sc.y;
	}
	struct RayInv { int unused; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3, const RayFlags &) { RayInv r; r.unused = 0; return r; }

	struct Objects { float bg1, bg2, lense, sphere, mirror, frame; vec3 mirror_pos; vec2 bg1_xz; };
	static SDF_HD float blob(vec3 p)
	{
		return lerp1(sd_sphere(p, 1.f), sd_box(p, V3s(1.f)), 0.65f) - 0.1f;
	}
	static SDF_HD Objects eval_objects(const FrameU &U, vec3 p)
	{
		Objects o;
		vec3 b1 = p - V3(0.f, -5.f, 0.f);
		vec2 r1 = op_rep_inf_c(V2(b1.x, b1.z), 3.f, 1.0f / 3.f);
		o.bg1_xz = r1;
		o.bg1 = blob(V3(r1.x, b1.y, r1.y));

		vec3 b2 = p - V3(0.f, 5.f, 0.f);
		vec2 r2 = op_rep_inf_c(V2(b2.x, b2.z), 10.f, 1.0f / 10.f);
		o.bg2 = blob(V3(r2.x, b2.y, r2.y));

		vec3 lp = abs(p);
		lp.z = lp.z - 5.1f;
		o.lense = max1(-sd_sphere(lp, 5.f), sd_sphere(p, 2.f));

		o.sphere = sd_sphere(p - V3(U.scene_var[SV_XPOS], U.scene_var[SV_YPOS], U.scene_var[SV_ZPOS]), 2.f);

		vec3 mp = p - V3(0.f, 0.f, -5.f);
		vec2 mr = rot2(V2(mp.x, mp.z), U.su[SU_MIRROR_S], U.su[SU_MIRROR_C]);
		mp = V3(mr.x, mp.y, mr.y);
		o.mirror_pos = mp;
		o.mirror = sd_box(mp, V3(1.f, 2.f, 0.1f));
		o.frame = sd_box(mp, V3(1.1f, 2.1f, 0.08f));
		return o;
	}
	// The mirror pane and its frame (two boxes, 45 of the 180 instructions of an evaluation) lie in
	// the ball of radius 2.38 about (0, 0, -5), whatever the pane's rotation about its vertical
	// axis; sd_box is an exact distance, so both are >= |p - c| - 2.38 and can be left out of the
	// min() whenever that bound is not below the running minimum (0.01 of slack for rounding;
	// checked numerically in tests/test_scene_bounds_cpu.py).
	//
	// The two blob fields are slabs about y = -5 and y = +5: a blob is lerp(sphere r 1, box 1, 0.65) - 0.1
	// of a point whose height above the field's plane is h, and both the sphere's and the box's distance
	// are >= |h| - 1, so is their blend (weights 0.35 / 0.65), hence  blob >= |h| - 1.1.  Lens, light ball
	// and pane live between the fields: close to them (nearer than ~3.9 minus the height) the repetition,
	// sphere and box of both fields (100 of the ~200 instructions) cannot lower the minimum and are left
	// out.  0.01 of slack; only below |y| = 1024, where the quantities are O(1000) at most.
	static SDF_HD float blob_field_lower_bound(float h) { return abs1(h) - 1.11f; }
	// skip an object that lies in the ball (c, radius) when that ball is not nearer than the running minimum d:
	// exact distances are >= |p - c| - radius (0.01 of slack, as for the pane)
	static SDF_HD bool ball_is_farther(vec3 p, vec3 c, float radius, float d)
	{
		const vec3 v = p - c;
		const float k = max1(d, 0.f) + (radius + 0.01f);
		return dot(v, v) >= k * k;
	}
	// Between the two blob fields (blob >= |h| - 1.11: nothing of them where |y| < 3.88) lie only the lens (in the ball of radius 2 about
	// the origin), the light ball (radius 2) and the pane with its frame (in the ball of radius 2.38 about (0, 0, -5)).  What a ray has
	// left to go -- a shadow ray ends at its light, and the extension lights hang 3 high -- meets nothing once it is in that slab for
	// good (it ends there, or leaves it through a gap of the upper field: below; and has crossed |y| = 3.86 if it started outside) and
	// has each ball behind it or passes it at a distance:
	// the distance along the ray from which that holds.  0.01 - 0.02 of slack for rounding; NaN or an
	// end outside the slab: never.
	static SDF_HD float ball_left_behind(vec3 start, vec3 dir, vec3 c, float radius)
	{
		const vec3 v = start - c;
		const float b = dot(v, dir), cc = dot(v, v) - radius * radius, dd = dot(dir, dir); // (dir need not be a unit vector: sdfr_lib.h)
		const float disc = b * b - cc * dd;
		if (disc < 0.f) return 0.f; // the line passes at a distance
		return (sqrt1(disc) - b) / dd + 0.01f; // the far intersection (NaN stays NaN)
	}
	static SDF_HD float escapes_from(const FrameU &U, vec3 start, vec3 dir, float range)
	{
		const float y_end = start.y + dir.y * range;
		float t = 0.f;
		if (!(abs1(y_end) < 3.86f))
		{
			// ... or it leaves through the upper field, whose blobs (within 1.11 of the centres of 10 x 10 cells) are mostly gaps, and there is
			// nothing above: while y runs from 3.86 to 6.14 the ray covers an interval of x and one of z; if either stays clear of every
			// centre's band, no blob is met.  (A ray that starts on a blob of that field starts inside both bands.)  Towards the sun: 7 in 10.
			if (!(dir.y > 1e-3f) || !(y_end > 6.2f)) return 3e38f;
			const float inv = 1.f / dir.y;
			const float ta = (max1(start.y, 3.86f) - start.y) * inv, tb = max1((6.14f - start.y) * inv, 0.f);
			const float xa = start.x + dir.x * ta, xb = start.x + dir.x * tb, za = start.z + dir.z * ta, zb = start.z + dir.z * tb;
			const float xl = (min1(xa, xb) - 1.13f) * 0.1f, xh = (max1(xa, xb) + 1.13f) * 0.1f, zl = (min1(za, zb) - 1.13f) * 0.1f, zh = (max1(za, zb) + 1.13f) * 0.1f;
			const bool clear_x = floor1(xl) == floor1(xh) && xl > floor1(xl), clear_z = floor1(zl) == floor1(zh) && zl > floor1(zl);
			if (!((clear_x || clear_z) && abs1(start.x) < 1e4f && abs1(start.z) < 1e4f)) return 3e38f;
			if (start.y < -3.86f) t = (-3.86f - start.y) * inv; // out of the lower field first
		}
		else if (!(abs1(start.y) < 3.86f)) t = ((start.y > 0.f ? 3.86f : -3.86f) - start.y) / dir.y; // y(t) = +-3.86: from there to the end inside the slab
		const vec3 ball = V3(U.scene_var[SV_XPOS], U.scene_var[SV_YPOS], U.scene_var[SV_ZPOS]);
		const float t0 = ball_left_behind(start, dir, V3(0.f, 0.f, 0.f), 2.02f), t1 = ball_left_behind(start, dir, ball, 2.02f), t2 = ball_left_behind(start, dir, V3(0.f, 0.f, -5.f), 2.40f);
		t = max1(max1(t, t0), max1(t1, t2));
		return (t == t && t0 == t0 && t1 == t1 && t2 == t2) ? t : 3e38f;
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &, vec3 p, vec3, bool)
	{
		// min() over the same objects as map(), any order gives the same bits.  Which order is cheap depends on where the
		// point is: close to a blob field (|y| > 2.9: the shadow rays creeping away from the blob they started on, a dozen
		// steps each) the field comes first and lens, light ball (two square roots each) and pane are skipped behind a
		// squared-distance test against their bounding balls; between the fields they come first and the fields are skipped
		// behind their slab bounds.
		const bool far_out = !(abs1(p.y) < 1024.f); // or NaN: no culling there
		float d = 3e38f;
		const bool near_a_field = abs1(p.y) > 2.9f && !far_out;
		if (near_a_field)
		{
			vec3 b1 = p - V3(0.f, -5.f, 0.f);
			if (!(blob_field_lower_bound(b1.y) >= d))
			{
				vec2 r1 = op_rep_inf_c(V2(b1.x, b1.z), 3.f, 1.0f / 3.f);
				d = min1(d, blob(V3(r1.x, b1.y, r1.y)));
			}
			vec3 b2 = p - V3(0.f, 5.f, 0.f);
			if (!(blob_field_lower_bound(b2.y) >= d))
			{
				vec2 r2 = op_rep_inf_c(V2(b2.x, b2.z), 10.f, 1.0f / 10.f);
				d = min1(d, blob(V3(r2.x, b2.y, r2.y)));
			}
		}
		// lens: cut out of the sphere of radius 2 about the origin, so inside that ball; light ball: a sphere of radius 2
		if (!near_a_field || !ball_is_farther(p, V3(0.f, 0.f, 0.f), 2.f, d))
		{
			vec3 lp = abs(p);
			lp.z = lp.z - 5.1f;
			d = min1(d, max1(-sd_sphere(lp, 5.f), sd_sphere(p, 2.f)));
		}
		const vec3 ball = V3(U.scene_var[SV_XPOS], U.scene_var[SV_YPOS], U.scene_var[SV_ZPOS]);
		if (!near_a_field || !ball_is_farther(p, ball, 2.f, d)) d = min1(d, sd_sphere(p - ball, 2.f));

		vec3 mp = p - V3(0.f, 0.f, -5.f);
		const float k = max1(d, 0.f) + (2.38f + 0.01f);
		if (!(dot(mp, mp) >= k * k))
		{
			vec2 mr = rot2(V2(mp.x, mp.z), U.su[SU_MIRROR_S], U.su[SU_MIRROR_C]);
			mp = V3(mr.x, mp.y, mr.y);
			d = min1(d, sd_box(mp, V3(1.f, 2.f, 0.1f)));
			d = min1(d, sd_box(mp, V3(1.1f, 2.1f, 0.08f)));
		}
		if (near_a_field) return d;

		vec3 b1 = p - V3(0.f, -5.f, 0.f);
		if (far_out || !(blob_field_lower_bound(b1.y) >= d))
		{
			vec2 r1 = op_rep_inf_c(V2(b1.x, b1.z), 3.f, 1.0f / 3.f);
			d = min1(d, blob(V3(r1.x, b1.y, r1.y)));
		}
		vec3 b2 = p - V3(0.f, 5.f, 0.f);
		if (far_out || !(blob_field_lower_bound(b2.y) >= d))
		{
			vec2 r2 = op_rep_inf_c(V2(b2.x, b2.z), 10.f, 1.0f / 10.f);
			d = min1(d, blob(V3(r2.x, b2.y, r2.y)));
		}
		return d;
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		Objects o = eval_objects(U, sp.pos);
		vec2 cell_index = (V2(sp.pos.x, sp.pos.z) - o.bg1_xz) / 3.f;
		if (on_surface(U, o.bg1))
		{
			vec2 a = cell_index * 0.3f;
			vec3 c1 = V3(sin1(a.x) * 0.5f + 0.5f, sin1(a.y) * 0.5f + 0.5f, 1.f);
			vec3 c2 = cell_index.x < 0.01f ? V3(0.f, 1.f, 0.f) : V3(0.f, 0.f, 1.f);
			vec3 c = lerp(c1, c2, 0.25f);
			m.diffuse = V4(c.x, c.y, c.z, 1.f);
			set_rgb(m.specular, 1.f);
			m.reflection = V3s(0.5f);
		}
		else if (on_surface(U, o.bg2))
		{
			m.diffuse = V4(1.f, 0.5f, 0.f, 1.f);
			set_rgb(m.specular, 1.f);
			m.reflection = V3s(0.5f);
		}
		else if (on_surface(U, o.lense))
		{
			m.diffuse = V4(0.3f, 0.3f, 0.3f, 1.f);
			m.refraction = V3(0.9f, 0.9f, 0.9f);
		}
		else if (on_surface(U, o.sphere))
		{
			m.diffuse = V4(1.f, 0.2f, 0.2f, 1.f);
			m.emissive = V3(8.f, 0.f, 0.f);
			set_rgb(m.specular, 1.f);
			m.reflection = V3s(0.25f);
		}
		else if (on_surface(U, o.mirror))
		{
			m.diffuse = V4(0.1f, 0.1f, 0.1f, 1.f);
			float mix = U.scene_var[SV_MIXING];
			m.refraction = V3s(mix);
			m.reflection = V3s(1.f - mix);
		}
		else if (on_surface(U, o.frame))
		{
			m.mpos = o.mirror_pos;
			m.id = MAT_WOOD;
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	// the scene carries its own copy of the sky (sdf_scene_lense.hlsl:108-117), same arithmetic
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneGems
{
	static const char *name() { return "gems"; }
	static constexpr int tile_w_log2 = 4; // 16 x 4 pixels per wave: configuration 5g 1.579 -> 1.517 ms (sdfr_render_pixel.h)
	static constexpr bool square_units = true; // the gems in the middle of the picture: with 8 lights and depth 4 their tiles render for 0.9 ms; first, not last: configuration 5g 1.657 -> 1.563 ms
	static constexpr bool persistent_tiles = true; // with 8 lights and depth 4 (configuration 5g) 2.88 -> 2.80 ms; the plain scene 1.10 -> 1.09
	static constexpr int retire_after = 1; // configuration 5g with tile rows 1.74 (8) / 1.68 (2) / 1.71 (1) ms, with the squares 1.480 (1) / 1.505 (2) / 1.510 (3) / 1.562 (4) / 1.81 (8) (profiles/r03_launch_experiments.txt)
	static constexpr int waves_per_simd = 7; // configuration 5g, with its escaped shadow rays delivered: 1.368 (5) / 1.336 (6) / 1.336 (7) / 1.536 (8) ms; the plain scene 0.799 / 0.818 / 0.779 / 0.873 (sdfr_pixel_kernel.h)
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static constexpr bool inline_escaped_shadows = true; // a floor pixel's shadow rays pass the ring at a distance (sdfr_pixel.h)
	static const char *variables() { return ""; }
	enum { SU_ROT_S = 0, SU_ROT_C = 1 };
	static SDF_HD void prepare(FrameU &U)
	{
		vec2 sc = sincos1(U.stime * 0.5f);
		U.su[SU_ROT_S] = sc.x;
		U.su[SU_ROT_C] = sc.y;
	}
	struct RayInv { GroundInv ground; bool rising; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		r.rising = dir.y >= 0.f;
		return r;
	}
	// gems() is a smooth maximum of three planes, never below any of them (the blend only adds): >= plane3 = p.y - 1.13.
	// Above y = 1.14 a ray that does not descend stays 0.01 clear of every gem, and the floor is behind it.
	// ... or its line passes the ring at a distance: a gem reaches 0.13 / 0.92 = 0.141 from its axis (plane2 and plane3 give
	// q.x <= q.y <= 0.13, the folds q.x >= 0.92 rho) and stands between y = 1 and 1.13 (plane2 with q.x >= 0: q.y >= 0), so
	// the ring lies in the ball about (0, 1.065, 0) of radius sqrt(1.141^2 + 0.065^2) = 1.143 (1.17: slack).
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 1.14f, V3(0.f, 1.065f, 0.f), 1.17f); }
	static SDF_HD float gems(const FrameU &U, vec3 p, float *ring_index)
	{
		vec2 xz = rot2(V2(p.x, p.z), U.su[SU_ROT_S], U.su[SU_ROT_C]);
		*ring_index = op_rep_angle(&xz, 8.f);
		vec3 q = V3(xz.x - 1.f, p.y - 1.f, xz.y);
		vec2 qxz = V2(q.x, q.z);
		op_rep_angle(&qxz, 8.f);
		q = V3(qxz.x, q.y, qxz.y);
		float plane1 = sd_plane(q - V3(0.1f, 0.1f, 0.f), V3(0.707f, 0.707f, 0.f));
		float plane2 = sd_plane(q, V3(0.707f, -0.707f, 0.f));
		float plane3 = sd_plane(q - V3(0.f, 0.13f, 0.f), V3(0.f, 1.f, 0.f));
		return op_smax2_c(op_smax2_c(plane1, plane2, 0.001f, 1.0f / 0.001f), plane3, 0.001f, 1.0f / 0.001f);
	}
	// A lower bound of gems() that needs neither of its two angular folds (an atan2, a sincos and a division each: 200 of
	// the 260 instructions of an evaluation).  The folds keep lengths and turn the point into a sector of +-pi/8 about
	// the local x axis, so q.x >= 0.92 rho >= 0, with rho the horizontal distance from the nearest gem's axis, itself
	// >= |r - 1| (the axes stand on the circle of radius 1, r = distance from the scene's axis).  gems() is a smooth
	// maximum of three planes and never below any of them:
	//   plane3 = p.y - 1.13;   plane2 = 0.707 (q.x - q.y) >= 0.707 (1 - p.y);
	//   max(plane1, plane2) >= (plane1 + plane2) / 2 = 0.707 (q.x - 0.1) >= 0.65 |r - 1| - 0.0707.
	// 0.01 of slack for rounding.  Where the bound is not below the floor's distance the gems cannot be the minimum: a ray
	// creeping away from the floor it started on (a dozen steps of every shadow ray towards the eight lights), anything
	// high above or far from the ring.  Checked numerically in tests/test_scene_bounds_cpu.py.
	static SDF_HD float gems_lower_bound(vec3 p)
	{
		const float r = length(V2(p.x, p.z));
		return max1(max1(p.y - 1.13f, 0.707f * (1.f - p.y)), 0.65f * abs1(r - 1.f) - 0.0707f) - 0.01f;
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		if (gems_lower_bound(p) >= d) return d;
		float idx;
		return min1(d, gems(U, p, &idx));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		if (gems_lower_bound(sp.pos) >= 2.f * U.dist_eps) return; // not on a gem
		float idx;
		if (on_surface(U, gems(U, sp.pos, &idx)))
		{
			bool ruby = frac1(idx * 0.5f + 0.25f) > 0.5f;
			m.diffuse.x = 0.8f;
			m.diffuse.y = ruby ? 0.1f : 0.7f;
			m.diffuse.z = ruby ? 0.3f : 0.1f;
			set_rgb(m.specular, 1.f);
			m.refraction = V3s(0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneLightShadows
{
	static const char *name() { return "light_shadows"; }
	static constexpr bool persistent_tiles = true; // expensive, uneven tiles: resident waves pulling tiles win (sdfr_render_pixel.h)
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return ""; }
	// su layout: 5 x {x, y_geometry, z, y_light} then 5 x rgb (colour of sphere i)
	enum { SU_SPHERES = 0, SU_COLORS = 20 };
	static SDF_HD vec3 color_from_index(uint32_t index)
	{
		float h = (float)index / 5.f;
		vec3 c = hsv_to_rgb(V3(h, 1.f, 1.f));
		return c / rgb_to_brightness(c);
	}
	static SDF_HD void prepare(FrameU &U)
	{
		float time = U.stime * 0.25f;
		for (uint32_t i = 0; i < 5; ++i)
		{
			time = time + SDFR_PI * 2.f / 5.f;
			float cx = cos1(time * 1.f) * -5.f;
			float c2 = cos1(time * 2.f);
			float sz = sin1(time * 2.f) * 2.f;
			U.su[SU_SPHERES + 4 * i + 0] = cx;
			U.su[SU_SPHERES + 4 * i + 1] = (c2 * -0.5f + 0.5f) * 2.f + 1.f;
			U.su[SU_SPHERES + 4 * i + 2] = sz;
			U.su[SU_SPHERES + 4 * i + 3] = (c2 * -0.5f + 0.5f) * 2.f + 0.5f;
			vec3 c = color_from_index(i);
			U.su[SU_COLORS + 3 * i + 0] = c.x;
			U.su[SU_COLORS + 3 * i + 1] = c.y;
			U.su[SU_COLORS + 3 * i + 2] = c.z;
		}
	}
	struct RayInv { GroundInv ground; bool is_shadow; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &f)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		r.is_shadow = f.is_shadow;
		return r;
	}
	// The rounded cubes end at y = 1.5 (centre 1, half size 0.4, rounded by 0.1), the light bulbs -- spheres of radius 0.2 whose centres bob
	// between y = 1 and 3, and which a shadow ray does not see at all -- at 3.2: a ray that does not descend (the floor is behind it) is
	// gone above that (0.02 of slack).
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &R, vec3 p, vec3 dir) { return dir.y >= 0.f && p.y > (R.is_shadow ? 1.52f : 3.22f); }
	static SDF_HD float cubes(vec3 p)
	{
		vec2 rep = op_rep_lim(V2(p.x, p.z), V2(2.f, 2.f), V2(3.f, 3.f));
		return sd_box(V3(rep.x, p.y, rep.y) - V3(0.f, 1.f, 0.f), V3s(0.4f)) - 0.1f;
	}
	static SDF_HD float sphere(const FrameU &U, vec3 p, int i)
	{
		return sd_sphere(p - V3(U.su[SU_SPHERES + 4 * i], U.su[SU_SPHERES + 4 * i + 1], U.su[SU_SPHERES + 4 * i + 2]), 0.2f);
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		if (!R.is_shadow) // the light bulbs do not shadow their own light
		{
#pragma unroll
			for (int i = 0; i < 5; ++i)
				d = min1(d, sphere(U, p, i));
		}
		return min1(d, cubes(p));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
#pragma unroll
		for (int i = 0; i < 5; ++i)
		{
			if (on_surface(U, sphere(U, sp.pos, i)))
				m.emissive = V3(U.su[SU_COLORS + 3 * i], U.su[SU_COLORS + 3 * i + 1], U.su[SU_COLORS + 3 * i + 2]);
		}
		if (on_surface(U, cubes(sp.pos)))
		{
			set_rgb(m.diffuse, 0.65f);
			set_rgb(m.specular, 0.75f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L)
	{
		if (i < 1 || i > 5) return false;
		int k = i - 1;
		L.pos = V3(U.su[SU_SPHERES + 4 * k], U.su[SU_SPHERES + 4 * k + 3], U.su[SU_SPHERES + 4 * k + 2]);
		L.directional = false;
		L.extend = 0.25f;
		L.falloff = 0.25f;
		L.color = V3(U.su[SU_COLORS + 3 * k], U.su[SU_COLORS + 3 * k + 1], U.su[SU_COLORS + 3 * k + 2]) * 0.5f;
		return true;
	}
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

} // namespace sdfr
