// sdfr_scenes3.h -- five more ahead-of-time scene functors (device code, host-compilable):
// fractal2, shell, spiral, terrain, tiling (Engine/shader/scenes/sdf_scene_<name>.hlsl).
#pragma once
#include "sdfr_scenes2.h"

namespace sdfr {

// =========================================================================================
struct SceneFractal2
{
	static const char *name() { return "fractal2"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return "VAR_slider(min = -5, max = 5, step = 0.01, start = 0)"; } // declared, unused by the scene
	enum { SU_SLICE_SHIFT = 0 };
	static SDF_HD void prepare(FrameU &U) { U.su[SU_SLICE_SHIFT] = U.stime * 0.5f; }
	struct RayInv { GroundInv ground; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		return r;
	}
	// the 6-level fold of `fractal` with every second ring of blocks pushed outwards
	static SDF_HD float fold(vec3 q)
	{
		const float size = 1.f;
		float d = 1e30f, scale = 1.f;
#pragma unroll
		for (int i = 0; i < 6; ++i)
		{
			d = min1(d, div_c(sd_box(q, V3s(size * 0.5f)), scale, 1.0f / scale)); // scale = 3^i: verified divisors
			q = abs(q);
			sort3_desc(q.y, q.x, q.z);
			q.y = q.y - size * 2.f / 3.f;
			q.z = q.z - step1(size * 0.5f / 3.f, q.z) * size / 3.f * 1.001f;
			q.y = q.y + size / 3.f;
			sort3_desc(q.y, q.x, q.z);
			q.y = q.y - size / 3.f;
			q = q * 3.f;
			scale = scale * 3.f;
		}
		return d;
	}
	static SDF_HD float wrap(float v, float lower, float upper)
	{
		const float range = upper - lower;
		const float reduced = (v - lower) / range;
		return (reduced - floor1(reduced)) * range + lower;
	}
	// Every box lies in the ball of radius 1.21 about (0, 1, 0): between two levels the point is folded (abs, sort: lengths kept), moved by
	// (-1/3, -0.3337) and then by 1/3 along an axis -- 0.472 + 0.333 = 0.805 at most -- and scaled by 3, so a point inside a box of level
	// i (|q_i| <= 0.866) has |q_(i-1)| <= |q_i| / 3 + 0.805: 1.094, 1.17, 1.195, ... < 1.2075.  fold() is a min() over exact box distances,
	// so it is >= |p - c| - 1.21: skipped where that is not below the floor's distance, and a ray that leaves the ball and the floor is a miss
	// (0.02 of slack; checked numerically in tests/test_scene_bounds_cpu.py).
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 2.24f, V3(0.f, 1.f, 0.f), 1.23f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		const vec3 v = p - V3(0.f, 1.f, 0.f);
		const float k = max1(d, 0.f) + 1.23f;
		if (dot(v, v) >= k * k) return d;
		return min1(d, fold(v));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		const vec3 base = sp.pos - V3(0.f, 1.f, 0.f);
		if (on_surface(U, fold(base)))
		{
			// a glowing slice sweeps diagonally through the fractal
			const float slice = dot(base, V3s(1.f));
			float diff = slice - U.su[SU_SLICE_SHIFT] - snoise3(base) * 0.5f;
			diff = wrap(diff, -0.5f, 0.5f);
			const float colorize = sat1(0.01f - abs1(diff)) / 0.01f;
			const float len = length(base);
			m.diffuse.x = 1.f;
			m.diffuse.y = 0.8f;
			m.diffuse.z = 0.1f;
			m.emissive = V3(0.8f, 0.3f, 0.1f) * colorize * 1.5f + V3(0.1f, 0.5f, 0.1f) * sat1((0.6f - len) * 10.f);
			set_rgb(m.specular, 0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L)
	{
		if (i != 0) return false;
		L.pos = V3(-1.f, -4.f, 2.f);
		L.directional = true;
		L.color = V3(1.f, 1.f, 1.f);
		L.extend = 0.f;
		L.falloff = 0.f;
		return true;
	}
	static SDF_HD float ambient() { return 0.1f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneShell
{
	static const char *name() { return "shell"; }
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
	// a cube turned into two nested shells, cut open along x = 0
	static SDF_HD float shells(vec3 p)
	{
		float c = sd_box(p - V3(0.f, 1.f, 0.f), V3s(0.5f));
		c = op_shell(c, 0.f, 0.3f);
		c = op_shell(c, -0.05f, 0.05f);
		return max1(c, -sd_plane(p, V3(-1.f, 0.f, 0.f)));
	}
	// op_shell(d, inner, outer) = |d - middle| - half >= d - outer, so shells() >= cube - 0.35 >= |p - (0, 1, 0)| - 0.866 - 0.35: the shells
	// lie in the ball of radius 1.2161 about the cube's centre (1.24: slack), below y = 2.24; a ray that leaves it and the floor is a miss
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 2.25f, V3(0.f, 1.f, 0.f), 1.24f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, shells(p));
		return min1(d, ground_dist(p, fast, R.ground)); // this scene lists the floor last
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		if (on_surface(U, shells(sp.pos)))
		{
			m.diffuse = V4(0.6f, 0.5f, 0.2f, 1.f);
			set_rgb(m.specular, 0.5f);
			m.reflection = V3s(0.15f);
		}
		else
			ground_material(U, sp, m);
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L)
	{
		if (!sun_light(i, L)) return false;
		L.color = V3(1.f, 1.2f, 1.f);
		return true;
	}
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color_mix(dir, U.sky_s, U.sky_c, 1.f, 0.f); }
};

// =========================================================================================
struct SceneSpiral
{
	static const char *name() { return "spiral"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return ""; }
	// a spring hopping along a parabola: everything about the hop is frame-uniform
	enum { SU_SHIFT_X = 0, SU_SPRING_S, SU_SPRING_C, SU_CENTER_Y, SU_LENGTH, SU_REACH };
	static SDF_HD void prepare(FrameU &U)
	{
		const float speed = 1.5f, width = 4.f, height = 6.f, pen = 2.f;
		float spring_length = 3.f;
		const float total_x = U.stime * speed;
		const float arc_pos = frac1(total_x / width);
		const float x = arc_pos * width;
		const float y = arc_pos * (1.f - arc_pos) * 4.f * height;
		const float y_top = y - pen + spring_length;
		const float y_bottom = max1(y - pen, 0.f);
		const float dydx = (1.f - 2.f * arc_pos) * 4.f * height / width;
		const float spring_angle = -atan1(dydx) - SDFR_PI * 0.5f;
		const vec2 sc = sincos1(spring_angle);
		U.su[SU_SHIFT_X] = x;
		U.su[SU_SPRING_S] = sc.x;
		U.su[SU_SPRING_C] = sc.y;
		U.su[SU_CENTER_Y] = (y_top + y_bottom) * 0.5f + 0.1f;
		U.su[SU_LENGTH] = y_top - y_bottom;
		// the spring's bounding ball about its middle (spring_lower_bound below: within 0.102 of the tube's circle radially, and within
		// pitch + 0.102 of its ends along the axis, nothing is nearer than 0.002)
		const float len = y_top - y_bottom, half = len * 0.5f + len / 4.5f + 0.102f;
		U.su[SU_REACH] = sqrt1(1.102f * 1.102f + half * half) + 0.02f;
	}
	struct RayInv { GroundInv ground; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		return r;
	}
	static SDF_HD float helix(vec3 pos, float r1, float h, float r2, float angle_start, float angle_end)
	{
		const float pitch = SDFR_TAU * h / (angle_end - angle_start);
		const float rel_height = pitch * atan21(pos.z, pos.x) / SDFR_TAU;
		const float start_height = pitch * angle_start / SDFR_TAU;
		float y = pos.y;
		float turn = clamp1(y, pitch * 0.5f, h - pitch * 0.5f);
		turn = turn - (rel_height - start_height);
		y = y - (rel_height - start_height);
		const float closest = rne1(turn / pitch) * pitch;
		const float axial = closest - y;
		const float radial = length(V2(pos.x, pos.z)) - r1;
		const float body = length(V2(radial, axial)) - r2;
		const vec2 s0 = sincos1(angle_start), s1 = sincos1(angle_end);
		const vec3 cap1 = V3(r1 * s0.y, 0.f, r1 * s0.x);
		const vec3 cap2 = V3(r1 * s1.y, h, r1 * s1.x);
		return min1(body, min1(length(pos - cap1) - r2, length(pos - cap2) - r2));
	}
	static SDF_HD float spring(const FrameU &U, vec3 p)
	{
		vec3 q = p;
		q.y = q.y - U.su[SU_CENTER_Y];
		const vec2 r = rot2(V2(q.x, q.y), U.su[SU_SPRING_S], U.su[SU_SPRING_C]);
		q = V3(r.x, r.y, q.z);
		const float len = U.su[SU_LENGTH];
		return helix(q + V3(0.f, len * 0.5f, 0.f), 1.f, len, 0.1f, 0.f, 4.5f * SDFR_TAU) * 0.98f;
	}
	// A lower bound of spring() without the helix (an atan2, a rounding, three lengths).  In the spring's frame, rho the distance from its
	// axis and y the height above its lower end: the tube's distance is length(rho - 1, axial) - 0.1 with |axial| >= dist(y, [0, h]) - pitch
	// (the nearest turn is sought within half a pitch of the clamped height, the helix's own offset is another half), and the caps sit
	// on that circle at y = 0 and y = h: everything is >= max(|rho - 1|, dist(y, [0, h]) - pitch) - 0.1, times the scene's 0.98.  0.01 of
	// slack.  Checked numerically in tests/test_scene_bounds_cpu.py.
	static SDF_HD float spring_lower_bound(const FrameU &U, vec3 p)
	{
		vec3 q = p;
		q.y = q.y - U.su[SU_CENTER_Y];
		const vec2 r = rot2(V2(q.x, q.y), U.su[SU_SPRING_S], U.su[SU_SPRING_C]);
		const float len = U.su[SU_LENGTH];
		const float y = r.y + len * 0.5f;
		const float beyond = max1(max1(-y, y - len), 0.f) - len / 4.5f;
		const float rho = length(V2(r.x, q.z));
		return (max1(abs1(rho - 1.f), beyond) - 0.1f) * 0.98f - 0.01f;
	}
	// floor + the spring, inside the ball of radius SU_REACH about its middle (0, SU_CENTER_Y, 0)
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir)
	{
		return ray_leaves_floor_and_ball(p, dir, U.su[SU_CENTER_Y] + U.su[SU_REACH], V3(0.f, U.su[SU_CENTER_Y], 0.f), U.su[SU_REACH]);
	}
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	// the floor scrolls under the spring: it is evaluated at x + shift
	static SDF_HD vec3 scrolled(const FrameU &U, vec3 p) { return V3(p.x + U.su[SU_SHIFT_X], p.y, p.z); }
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(scrolled(U, p), fast, R.ground));
		if (spring_lower_bound(U, p) >= d) return d;
		return min1(d, spring(U, p));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		SurfacePoint moved = sp;
		moved.pos = scrolled(U, sp.pos);
		ground_material(U, moved, m);
		if (on_surface(U, spring(U, sp.pos))) set_rgb(m.diffuse, 0.5f);
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneTerrain
{
	static const char *name() { return "terrain"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return "VAR_levels(min=1, max=10, step=1, start=2)"; }
	enum { SU_ROT_S = 0, SU_ROT_C = 1 };
	static SDF_HD void prepare(FrameU &U)
	{
		const vec2 sc = sincos1(1.f); // the fixed twist between octaves
		U.su[SU_ROT_S] = sc.x;
		U.su[SU_ROT_C] = sc.y;
	}
	struct RayInv { bool rising; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &) { RayInv r; r.rising = dir.y >= 0.f; return r; }
	// shape() = max(terrain, box of half size 5 about the origin) >= the box's distance >= p.y - 5: nothing above y = 5,
	// and this scene has no other object (no floor either)
	// The scene has no floor: a ray whose line passes the box's circumscribed ball (radius sqrt(75) = 8.66 about the origin;
	// 8.7) at a distance, or has it behind, is gone whichever way it points.
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &R, vec3 p, vec3 dir)
	{
		// ... and the terrain itself ends far below the box's lid: an octave lowers the running distance by 0.175 s at most (smax(n, d - 0.1 s)
		// is >= d - 0.1 s, and the polynomial smin() stays within a quarter of its 0.3 s below the smaller operand), so fbm(p, p.y) >=
		// p.y - 0.175 (1 + 1/2 + 1/4 + ...) = p.y - 0.35 whatever the number of levels: nothing above y = 0.35, and a ray that
		// does not descend is gone from 0.37 on (0.02 above it: dist_eps is 1e-3 at most).  Checked numerically in tests/test_scene_bounds_cpu.py.
		if (R.rising && p.y > 0.37f) return true;
		// ... or beyond any other face of the box and not coming back (shape() >= the box's distance >= |p_i| - 5 on every axis)
		if ((abs1(p.x) > 5.01f && p.x * dir.x >= 0.f) || (abs1(p.z) > 5.01f && p.z * dir.z >= 0.f) || (p.y < -5.01f && dir.y <= 0.f)) return true;
		return ray_passes_ball(p, dir, V3s(0.f), 8.7f);
	}
	static SDF_HD float lattice_noise(vec3 p) { return frac1(sin1(dot(p, V3(12.9898f, 78.233f, 34.531247f))) * 43758.5453f); }
	static SDF_HD float corner_sphere(vec3 cell, vec3 p, vec3 off)
	{
		const vec3 c = cell + off;
		return sd_sphere(p - c, lerp1(0.f, 0.3f, lattice_noise(c)));
	}
	// spheres of random radius on the 8 corners of the lattice cell
	static SDF_HD float base(vec3 p)
	{
		const vec3 cell = floor(p);
#if defined(__HIP_DEVICE_COMPILE__)
		// the radii hang on the cell alone (a sine hash per corner: two thirds of base()); where the whole wave stands
		// in one cell -- the coarse octaves, mostly -- eight lanes work out one radius each (WaveShare, sdfr_lib.h)
		uint32_t rank;
		if (WaveShare::agree(cell.x, cell.y, cell.z, 8u, &rank))
		{
			float *slots = WaveShare::slots();
			if (rank < 8u) slots[rank] = lerp1(0.f, 0.3f, lattice_noise(cell + V3((float)(rank >> 2), (float)((rank >> 1) & 1u), (float)(rank & 1u))));
			WaveShare::publish();
			float r[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) r[k] = slots[k];
			WaveShare::release();
			float nearest = 3e38f;
			// the same eight values as below, and min() over them in the same grouping
			float v[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) v[k] = sd_sphere(p - (cell + V3((float)(k >> 2), (float)((k >> 1) & 1), (float)(k & 1))), r[k]);
			nearest = min1(min1(min1(v[0], v[1]), min1(v[2], v[3])), min1(min1(v[4], v[5]), min1(v[6], v[7])));
			return nearest;
		}
#endif
		const float a = corner_sphere(cell, p, V3(0.f, 0.f, 0.f)), b = corner_sphere(cell, p, V3(0.f, 0.f, 1.f));
		const float c = corner_sphere(cell, p, V3(0.f, 1.f, 0.f)), d = corner_sphere(cell, p, V3(0.f, 1.f, 1.f));
		const float e = corner_sphere(cell, p, V3(1.f, 0.f, 0.f)), f = corner_sphere(cell, p, V3(1.f, 0.f, 1.f));
		const float g = corner_sphere(cell, p, V3(1.f, 1.f, 0.f)), h = corner_sphere(cell, p, V3(1.f, 1.f, 1.f));
		return min1(min1(min1(a, b), min1(c, d)), min1(min1(e, f), min1(g, h)));
	}
	// One octave: n = s * base(p) is blended into the running distance d by
	//     n' = smax(n, d - 0.1 s, 0.3 s);   d' = smin(n', d, 0.3 s).
	// base() -- eight corner spheres with hashed radii, eight sin1: 85 % of an octave -- is the distance
	// to the nearest of the eight corner spheres of the unit cell around p; some corner is no farther than
	// half the cell's diagonal and the radii are >= 0, so base(p) <= 0.8661 and n <= 0.8661 s.  op_smax2(a, b,
	// k) returns b, whatever a is, once b - a >= k (its blend weight h clamps to 0).  So where
	// d >= 1.3 s (> 0.8661 s + 0.1 s + 0.3 s) the octave's result does not depend on base(p) at all and any
	// stand-in a with b - a >= k gives the bits the real one would: high above the terrain an evaluation
	// costs a few dozen instructions instead of ~450.  (Checked in tests/test_scene_bounds_cpu.py.)
	static SDF_HD bool octave_needs_base(float d, float s) { return !(d >= 1.3f * s); }
	static SDF_HD float fbm(const FrameU &U, vec3 p, float d)
	{
		const vec3 r0 = V3(0.00f, 1.60f, 1.20f), r1 = V3(-1.60f, 0.72f, -0.96f), r2 = V3(-1.20f, -0.96f, 1.28f);
		float s = 1.0f;
		const int levels = ftoi1(U.scene_var[0]);
		for (int i = 0; i < levels; i++)
		{
			const float b = d - 0.1f * s;
			float n = b - 0.6f * s; // stand-in with b - n >= 0.3 s
			if (octave_needs_base(d, s)) n = s * base(p);
			n = op_smax2(n, b, 0.3f * s);
			d = op_smin(n, d, 0.3f * s);
			p = V3(dot(r0, p), dot(r1, p), dot(r2, p));
			const vec2 r = rot2(V2(p.x, p.z), U.su[SU_ROT_S], U.su[SU_ROT_C]);
			p = V3(r.x, p.y, r.y);
			s = 0.5f * s;
		}
		return d;
	}
	static SDF_HD float shape(const FrameU &U, vec3 p)
	{
		const float box = sd_box(p, V3(5.f, 5.f, 5.f));
		const float plane = sd_plane(p, V3(0.f, 1.f, 0.f));
		return max1(fbm(U, p, plane), box);
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &, vec3 p, vec3, bool) { return min1(3e38f, shape(U, p)); }
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		if (on_surface(U, shape(U, sp.pos)))
		{
			m.diffuse.x = 0.8f;
			m.diffuse.y = 0.8f;
			m.diffuse.z = 0.8f;
			set_rgb(m.specular, 0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneTiling
{
	static const char *name() { return "tiling"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	enum { V_M1 = 0, V_M2, V_WIDTH, V_RUN_LENGTH, V_RUN_FLIP, V_FLIP_CHANCE, V_TRUCHET_WIDTH };
	static const char *variables()
	{
		return "VAR_m1(min = -1, max = 3, step = 0.1, start = 1) VAR_m2(min = -1, max = 3, step = 0.1, start = 0) "
			   "VAR_width(min = 0.1, max = 0.5, step = 0.05, start = 0.4) VAR_run_length(min = 1, max = 10, step = 1, start = 4) "
			   "VAR_run_flip(min = 1, max = 10, step = 1, start = 2) VAR_flip_chance(min = 0, max = 1, steps = 0.05) "
			   "VAR_truchet_width(min = 0, max = 0.2, step = 0.01)";
	}
	enum { SU_PULSE = 0 };
	static SDF_HD void prepare(FrameU &U) { U.su[SU_PULSE] = U.stime * 2.f; }
	struct RayInv { GroundInv ground; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		return r;
	}
	// panes (centre height 4, half height 2) and cable (a cylinder of half height 2 about y = 4) end at y = 6: a ray above that which does
	// not descend has them and the floor behind it
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return dir.y >= 0.f && p.y > 6.02f; }
	static SDF_HD float pane(vec3 p, float x) { return sd_box(p - V3(x, 4.f, 0.f), V3(1.f, 2.f, 0.05f)); }
	static SDF_HD float cable(vec3 p) { return sd_capped_cylinder(p - V3(4.f, 4.f, 4.f), 2.f, 0.1f); }
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, pane(p, -4.f));
		d = min1(d, pane(p, 0.f));
		d = min1(d, pane(p, 4.f));
		d = min1(d, cable(p));
		return min1(d, ground_dist(p, fast, R.ground));
	}
	static SDF_HD vec3 cell_color(vec2 cell)
	{
		const float r = pcg_hashf((uint32_t)ftoi1(cell.x + cell.y * 217.743f));
		const float g = pcg_hashf(cell_hash_key(cell, 2475.235f));
		const float b = pcg_hashf(cell_hash_key(cell, 824.213f));
		return V3(r, g, b) / max1(max1(r, g), b);
	}
	static SDF_HD float weave_grey(const FrameU &U, vec2 uv)
	{
		const float m1 = U.scene_var[V_M1], m2 = U.scene_var[V_M2];
		uv = V2(uv.x * m1 + uv.y * m2, uv.x * m2 + uv.y * m1);
		const vec4 pattern = braid(uv, U.scene_var[V_WIDTH], U.scene_var[V_RUN_LENGTH], U.scene_var[V_RUN_FLIP], V2(-2.f, 0.f));
		const float grey = step1(-1.f, pattern.z) * (cos1(pattern.z * 50.f) * 0.5f + 0.5f);
		return grey * grey;
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		const vec2 uv = V2(sp.pos.x, sp.pos.y);
		if (on_surface(U, pane(sp.pos, -4.f)))
		{
			const vec4 v = voronoi(uv * 5.f, 0.45f);
			const vec3 c = v.w > 0.05f ? cell_color(V2(v.x, v.y)) * 1.1f : V3(0.25f, 0.25f, 0.25f);
			m.diffuse = V4(c.x, c.y, c.z, 1.f);
			set_rgb(m.specular, 0.4f);
			m.specular.w = 20.f;
		}
		else if (on_surface(U, pane(sp.pos, 0.f)))
		{
			const vec4 t = truchet_band(op_ab2uv(uv) * 3.f, U.scene_var[V_FLIP_CHANCE], U.scene_var[V_TRUCHET_WIDTH], V2(0.f, -1.f));
			const float green = t.w < 0.f ? 0.f : sin1(t.w * 2.f * SDFR_PI * 5.f + U.su[SU_PULSE]) * 0.5f + 0.5f;
			m.diffuse = V4(0.f, green * green, 0.f, 1.f);
		}
		else if (on_surface(U, pane(sp.pos, 4.f)))
		{
			const float g = weave_grey(U, op_ab2uv(uv * 5.f));
			m.diffuse = V4(g * 0.8f, g * 0.8f, g * 0.8f, 1.f);
		}
		else if (on_surface(U, cable(sp.pos)))
		{
			const vec3 cp = sp.pos - V3(4.f, 4.f, 4.f);
			const float angle = atan21(cp.z, cp.x);
			const float g = weave_grey(U, op_ab2uv(V2(angle * 0.1f, cp.y) * 8.f));
			m.diffuse = V4(g * 0.9f, g * 0.9f, g * 0.9f, 1.f);
		}
		else
		{
			ground_material(U, sp, m);
			if (on_surface(U, dot(sp.pos, V3(0.f, 1.f, 0.f)))) set_rgb(m.specular, 0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L)
	{
		if (!sun_light(i, L)) return false;
		L.color = V3(1.f, 1.2f, 1.f);
		return true;
	}
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color_mix(dir, U.sky_s, U.sky_c, 0.8f, 0.2f); }
};

} // namespace sdfr
