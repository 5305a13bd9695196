// sdfr_scenes2.h -- nine more ahead-of-time scene functors (device code, host-compilable):
// cube, gyroid, basic_transparency, basic_clouds, coordinate_material, distortion, table,
// sierpinski, neon (Engine/shader/scenes/sdf_scene_<name>.hlsl).  Same functor interface
// as sdfr_scenes.h.
#pragma once
#include "sdfr_scenes.h"

namespace sdfr {

// =========================================================================================
struct SceneCube
{
	static const char *name() { return "cube"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	enum { V_SIZE = 0, V_X, V_Y, V_Z, V_RED, V_GREEN, V_BLUE };
	static const char *variables()
	{
		return "VAR_size(min = 0.2, max = 2, start = 1, step = 0.2) VAR_xpos(min = -2, max = 2, start = 0, step = 0.1) "
			   "VAR_ypos(min = -2, max = 2, start = 0, step = 0.1) VAR_zpos(min = -2, max = 2, start = 0, step = 0.1) "
			   "VAR_red(min = 0, max = 1, start = 0.9, step = 0.05) VAR_green(min = 0, max = 1, start = 0.7, step = 0.05) "
			   "VAR_blue(min = 0, max = 1, start = 0.2, step = 0.05)";
	}
	static SDF_HD void prepare(FrameU &) {}
	struct RayInv { GroundInv ground; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		return r;
	}
	static SDF_HD float cube(const FrameU &U, vec3 p)
	{
		const vec3 c = V3(U.scene_var[V_X], U.scene_var[V_Y], U.scene_var[V_Z]);
		return sd_box(p - V3(0.f, 1.f, 0.f) - c, V3s(U.scene_var[V_SIZE]));
	}
	// floor + one box of half size `size` about (x, 1 + y, z): inside the ball of radius sqrt(3) size about its centre
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir)
	{
		const float size = abs1(U.scene_var[V_SIZE]);
		const vec3 c = V3(U.scene_var[V_X], 1.f + U.scene_var[V_Y], U.scene_var[V_Z]);
		return ray_leaves_floor_and_ball(p, dir, c.y + size * 1.001f + 0.01f, c, size * 1.7330f + 0.02f);
	}
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		return min1(d, cube(U, p));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		if (on_surface(U, cube(U, sp.pos)))
		{
			m.diffuse = V4(U.scene_var[V_RED], U.scene_var[V_GREEN], U.scene_var[V_BLUE], 1.f);
			set_rgb(m.specular, 0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneGyroid
{
	static const char *name() { return "gyroid"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return ""; }
	static SDF_HD void prepare(FrameU &) {}
	struct RayInv { int unused; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3, const RayFlags &) { RayInv r; r.unused = 0; return r; }
	// shape() = max(gyroid, cube) >= the cube's distance, and the scene has nothing else (no floor): a ray beyond a face of the cube
	// that does not come back, or whose line passes the cube's ball (radius sqrt(3) = 1.733; 1.75) at a distance, has left it
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir)
	{
		if ((abs1(p.x) > 1.01f && p.x * dir.x >= 0.f) || (abs1(p.y) > 1.01f && p.y * dir.y >= 0.f) || (abs1(p.z) > 1.01f && p.z * dir.z >= 0.f)) return true;
		return ray_passes_ball(p, dir, V3s(0.f), 1.75f);
	}
	// a gyroid shell clipped to the unit cube; no floor in this scene.  The shell's term is |sin x cos z + sin y cos x + sin z cos y| / 14
	// - 0.01 <= 3 / 14 - 0.01 = 0.2043: from 0.22 off the cube max() returns the cube's distance, and the three sincos are left out.
	static SDF_HD float shape(vec3 p)
	{
		const float box = sd_box(p, V3(1.f, 1.f, 1.f));
		if (box >= 0.22f) return box;
		const vec3 q = p * 7.f;
		const vec2 sx = sincos1(q.x), sy = sincos1(q.y), sz = sincos1(q.z);
		float g = dot(V3(sx.x, sy.x, sz.x), V3(sz.y, sx.y, sy.y)) / 14.f;
		g = abs1(g) - 0.01f;
		return max1(g, box);
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &, vec3 p, vec3, bool) { return min1(3e38f, shape(p)); }
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		if (on_surface(U, shape(sp.pos)))
		{
			m.diffuse.x = 0.9f;
			m.diffuse.y = 0.7f;
			m.diffuse.z = 0.2f;
			set_rgb(m.specular, 0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneBasicTransparency
{
	static const char *name() { return "basic_transparency"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return ""; }
	static SDF_HD void prepare(FrameU &) {}
	static SDF_HD float pane(vec3 p, float z) { return sd_box(p - V3(0.f, 2.f, z), V3(1.f, 1.f, 0.1f)); }
	// a ray continuing through a pane ignores the pane it just left (OBJECT_TRANSPARENT)
	struct RayInv { GroundInv ground; bool skip1, skip2, skip3; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &f)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		r.skip1 = f.has_transparent && pane(f.last_transparent_pos, -1.f) < U.dist_eps;
		r.skip2 = f.has_transparent && pane(f.last_transparent_pos, 0.f) < U.dist_eps;
		r.skip3 = f.has_transparent && pane(f.last_transparent_pos, 1.f) < U.dist_eps;
		return r;
	}
	// floor + three panes of half size (1, 1, 0.1) about (0, 2, -1 / 0 / 1): below y = 3, inside the ball of radius
	// sqrt(1 + 1 + 1.1^2) = 1.79 about (0, 2, 0)
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 3.01f, V3(0.f, 2.f, 0.f), 1.82f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		const float b1 = pane(p, -1.f), b2 = pane(p, 0.f), b3 = pane(p, 1.f);
		d = R.skip1 ? d : min1(d, b1);
		d = R.skip2 ? d : min1(d, b2);
		d = R.skip3 ? d : min1(d, b3);
		return d;
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		if (on_surface(U, pane(sp.pos, -1.f)))
			m.diffuse = V4(0.9f, 0.9f, 0.f, 0.3f);
		else if (on_surface(U, pane(sp.pos, 0.f)))
			m.diffuse = V4(0.f, 0.9f, 0.9f, 0.3f);
		else if (on_surface(U, pane(sp.pos, 1.f)))
			m.diffuse = V4(0.9f, 0.f, 0.9f, 0.3f);
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneBasicClouds
{
	static const char *name() { return "basic_clouds"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return "VAR_offset(min = -5, max = 5, step = 0.05)"; }
	static SDF_HD void prepare(FrameU &) {}
	static SDF_HD float slab(vec3 p) { return sd_box(p - V3(0.f, 5.f, 0.f), V3(2.f, 0.5f, 2.f)); }
	struct RayInv { GroundInv ground; bool skip_cloud; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &f)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		r.skip_cloud = f.has_transparent && slab(f.last_transparent_pos) < U.dist_eps;
		return r;
	}
	// floor + the cloud's slab, half size (2, 0.5, 2) about (0, 5, 0): below y = 5.5, inside the ball of radius sqrt(8.25) = 2.873 about its centre
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 5.52f, V3(0.f, 5.f, 0.f), 2.9f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		const float c = slab(p);
		return R.skip_cloud ? d : min1(d, c);
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		if (on_surface(U, slab(sp.pos)))
		{
			// density sampled at five points along the view ray
			const vec3 cloud_pos = sp.pos - V3(0.f, 5.f, 0.f);
			float thickness = 0.f + U.scene_var[0];
			for (int i = 0; i < 5; ++i)
				thickness = thickness + turbulence3(cloud_pos + sp.dir * 0.5f * (float)i);
			thickness = sat1(thickness);
			const float c = 1.f - thickness * 0.2f;
			m.diffuse = V4(c, c, c, thickness);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L)
	{
		if (i != 0) return false;
		L.pos = V3(-1.f, -4.f, 1.5f);
		L.directional = true;
		L.color = V3(1.f, 1.f, 1.f);
		L.extend = 0.f;
		L.falloff = 0.f;
		return true;
	}
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneCoordinateMaterial
{
	static const char *name() { return "coordinate_material"; }
	static const char *variables()
	{
		return "VAR_boxoffset(min = 0, max = 2, step = 0.1, start = 2) VAR_spherical(min = 0, max = 1, step = 1, start = 0) "
			   "VAR_thres(min=0,max=1,step=0.05, start=0.4)";
	}
	static SDF_HD void prepare(FrameU &) {}
	struct RayInv { GroundInv ground; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		return r;
	}
	static SDF_HD float shape(const FrameU &U, vec3 p)
	{
		const float sphere = sd_sphere(p - V3(0.f, 2.f, 0.f), 2.f);
		const float box = sd_box(p - V3(-1.f, 3.f + U.scene_var[0], -1.f), V3s(1.f));
		return max1(sphere, -box);
	}
	// floor + a sphere of radius 2 about (0, 2, 0) with a box cut out of it (max(sphere, -box) >= sphere)
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 4.01f, V3(0.f, 2.f, 0.f), 2.02f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		return min1(d, shape(U, p));
	}
	static SDF_HD vec3 to_spherical(vec3 p)
	{
		return V3(length(p), atan21(p.y, length(V2(p.x, p.z))), atan21(p.z, p.x));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		if (on_surface(U, shape(U, sp.pos)))
		{
			vec3 pos = (sp.pos - V3(0.f, 2.f, 0.f)) * 2.f;
			vec3 norm = sp.normal;
			if (U.scene_var[1] > 0.5f)
			{
				// grid in (r, theta, phi); the normal is transformed by a finite difference
				const vec3 sph = to_spherical(pos);
				const vec3 off = to_spherical(pos + norm * 0.01f);
				norm = normalize(off - sph);
				pos = sph * V3(1.f, 8.f / SDFR_PI, 8.f / SDFR_PI);
			}
			const float sel = mat_coordinate_grid(pos, norm, 0.02f);
			const bool on_line = sel > U.scene_var[2];
			m.diffuse.x = on_line ? 1.f : 0.8f;
			m.diffuse.y = on_line ? 0.f : 0.8f;
			m.diffuse.z = on_line ? 0.f : 0.8f;
			set_rgb(m.specular, 0.25f);
			m.specular.w = 100.f;
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneDistortion
{
	static const char *name() { return "distortion"; }
	static constexpr int waves_per_simd = 8; // 4.55 -> 4.40 ms at 4K (sdfr_pixel_kernel.h)
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
	// The displaced wall stays within 0.026 of its box (wall_lower_bound below): the box of half size (1, 1, 0.1) about
	// (0, 1.5, 0) lies in the ball of radius 1.418 about its centre, the wall in the one of radius 1.45, below y = 2.53.
	// A ray that does not descend (the floor is behind it) and is above that height, or whose line passes that ball at a
	// distance or has it behind, has nothing left to hit.
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 2.6f, V3(0.f, 1.5f, 0.f), 1.5f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	// displace a distance field by a height function with known Lipschitz bound
	static SDF_HD float distort(float obj, float val, float lip, float h)
	{
		const float actual = (obj - val) / sqrt1(1.f + lip * lip);
		return lerp1(actual, obj - h, sat1(obj / h - 1.f));
	}
	struct Wall { float box, val, noise; };
	static SDF_HD Wall wall(vec3 p)
	{
		Wall w;
		const vec3 bp = p - V3(0.f, 1.5f, 0.f);
		const float scaled = bp.y * 2.5f - 1.25f;
		const float index = scaled - floor1(scaled);
		const float offset = step1(0.5f, index);
		const float val_x = 1.f - pow1(sat1(sin1((bp.x + offset * 0.2f) * SDFR_PI * 5.f)), 10.f);
		const float val_y = 1.f - pow1(abs1(sin1(bp.y * SDFR_PI * 5.f)), 10.f);
		w.noise = turbulence3(bp * 7.5f);
		float v = min1(val_x, val_y);
		v = lerp1(v * 0.8f, v, w.noise);
		w.val = v;
		const float height = 0.025f, lip = 2.f;
		w.box = distort(sd_box(bp, V3(1.f, 1.f, 0.1f)), v * height, lip * height, height);
		return w;
	}
	// Away from the wall the displaced distance is the plain box distance minus the displacement
	// height: distort() blends to `obj - h` with weight sat(obj / h - 1) = 1 once obj >= 2 h, and
	// lerp1(a, b, 1) = round(round(b - a) + a) is b to within an ulp or two of obj.  So for
	// obj >= 0.05 the wall's distance is >= (obj - 0.026) * 0.9999, and the displacement pattern --
	// a turbulence, two sines, two pows: ~900 of the ~950 instructions of an evaluation -- need not
	// be evaluated when that bound cannot change the min() (or the material test).
	static SDF_HD float wall_lower_bound(vec3 p, bool *valid)
	{
		const float obj = sd_box(p - V3(0.f, 1.5f, 0.f), V3(1.f, 1.f, 0.1f));
		*valid = obj >= 0.05f;
		return (obj - 0.026f) * 0.9999f;
	}
	// More than a bound: from obj = 0.0625 on the displaced distance IS `obj - h`, bit for bit.  distort() returns
	// fma(t, b - a, a) with a = (obj - val) / 1.00125, b = obj - h, t = sat(obj / h - 1) = 1 (obj / h >= 2.5).  The pattern
	// value is in [0, 0.025] (v in [0, 1], the turbulence in [-1, 1]: v * (0.8 + 0.2 n)), so b / 2 <= a <= 2 b
	// (a <= obj / 1.00125 <= 2 obj - 0.05 from obj = 0.04994 on) and the subtraction b - a is exact (Sterbenz): the fma
	// rounds the real number b, a float already.  Only the last step or two of a ray that reaches the wall evaluate the
	// pattern.  Up to 1024 only: far beyond, the pattern's arguments overflow.  Bits compared in tests/test_scene_bounds_cpu.py.
	static SDF_HD bool wall_is_plain_box(float obj) { return obj >= 0.0625f && obj <= 1024.f; }
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		const float obj = sd_box(p - V3(0.f, 1.5f, 0.f), V3(1.f, 1.f, 0.1f));
		if (wall_is_plain_box(obj)) return min1(d, obj - 0.025f);
		if (obj >= 0.05f && (obj - 0.026f) * 0.9999f >= d) return d;
		return min1(d, wall(p).box);
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		bool valid;
		const float lb = wall_lower_bound(sp.pos, &valid);
		if (valid && lb >= 0.1f + 2.f * U.dist_eps) return; // (box - 0.1) < eps is impossible
		const Wall w = wall(sp.pos);
		if ((w.box - 0.1f) < U.dist_eps)
		{
			const vec3 brick = lerp(V3(0.5f, 0.1f, 0.1f), V3(0.8f, 0.2f, 0.2f), w.noise);
			const vec3 c = w.val < 0.15f ? V3(0.5f, 0.5f, 0.5f) : brick;
			m.diffuse.x = c.x;
			m.diffuse.y = c.y;
			m.diffuse.z = c.z;
			set_rgb(m.specular, 0.125f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneTable
{
	static const char *name() { return "table"; }
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
	struct Objects { float plate, legs, vase; vec3 p; };
	static SDF_HD Objects eval_objects(vec3 pos)
	{
		const float leg_width = 0.05f, leg_height = 0.7f, leg_distance = 1.f, plate_size = 1.2f, plate_height = 0.0175f;
		Objects o;
		vec3 p = pos;
		p.y = p.y - 0.4f;
		o.p = p;
		o.legs = sd_box(abs(p) - V3(leg_distance, leg_height * 0.5f, leg_distance), V3(leg_width, leg_height * 0.5f, leg_width));
		o.plate = sd_box(p - V3(0.f, leg_height + 0.025f, 0.f), V3(plate_size, plate_height, plate_size)) - 0.025f;
		// a vase of three smoothly merged spheres, hollowed out
		const float s1 = sd_sphere(p - V3(0.f, leg_height + 0.15f, 0.f), 0.2f);
		const float s2 = sd_sphere(p - V3(0.f, leg_height + 0.45f, 0.f), 0.17f);
		const float s3 = sd_sphere(p - V3(0.f, leg_height + 0.72f, 0.f), 0.15f);
		const float cut_top = sd_plane(p - V3(0.f, leg_height + 0.615f, 0.f), V3(0.f, 1.f, 0.f));
		const float cut_low = sd_plane(p - V3(0.f, leg_height + 0.1f, 0.f), V3(0.f, -1.f, 0.f));
		const float body = max1(op_smin_c(op_smin_c(s1, s2, 0.05f, 1.0f / 0.05f), s3, 0.025f, 1.0f / 0.025f), cut_low);
		o.vase = max1(max1(body, cut_top), -body - 0.01f);
		return o;
	}
	// Legs (|x|, |z| = 1 +- 0.05, up to y = 1.1, down to -0.3), plate (half size 1.2 rounded by 0.025, about y = 1.125) and vase
	// (within 0.22 of the axis, cut off at y = 1.715: vase >= cut_top) lie below y = 1.72 and in the ball of radius 1.80
	// about (0, 0.7, 0) (plate corner and leg foot are its farthest points).
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 1.73f, V3(0.f, 0.7f, 0.f), 1.85f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		const Objects o = eval_objects(p);
		d = min1(d, o.plate);
		d = min1(d, o.legs);
		return min1(d, o.vase);
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		const Objects o = eval_objects(sp.pos);
		if (on_surface(U, o.plate))
		{
			const float plank = floor1((o.p.x + 1.25f) * 4.f) / 8.f;
			m.mpos = o.p + V3(o.p.z * 0.2f, plank, 0.f);
			m.id = MAT_WOOD;
		}
		else if (on_surface(U, o.legs))
		{
			m.mpos = V3(o.p.x, o.p.z, o.p.y);
			m.id = MAT_WOOD;
		}
		else if (on_surface(U, o.vase))
		{
			m.mpos = o.p * 4.f;
			m.id = MAT_MARBLE_DARK;
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneSierpinski
{
	static const char *name() { return "sierpinski"; }
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
	// fold towards the nearest of four tetrahedron vertices, ten times
	static SDF_HD float tetra(vec3 p)
	{
		const vec3 a1 = V3(0.f, 1.f, 0.f), a2 = V3(-0.7f, 0.f, -0.5f), a3 = V3(0.7f, 0.f, -0.5f), a4 = V3(0.f, 0.f, 0.7f);
		const float scale = 2.f;
#pragma unroll
		for (int it = 0; it < 10; ++it)
		{
			vec3 c = a1;
			float best = length(p - a1);
			float d = length(p - a2);
			if (d < best) { c = a2; best = d; }
			d = length(p - a3);
			if (d < best) { c = a3; best = d; }
			d = length(p - a4);
			if (d < best) { c = a4; best = d; }
			p = scale * p - c * (scale - 1.f);
		}
		return length(p) / pow1(scale, 10.f) - 0.002f;
	}
	// Each fold doubles the distance from the chosen vertex, and the vertices lie within 1 of the origin: |2 p - c| >=
	// 2 |p| - 1, so |p_k| - 1 >= 2^k (|p| - 1) and tetra(p) >= |p| - 1.001: the gasket lies in the ball of radius 1.001
	// about (0, 1, 0), below y = 2.001.
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir) { return ray_leaves_floor_and_ball(p, dir, 2.01f, V3(0.f, 1.f, 0.f), 1.02f); }
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		// tetra(p) >= |p| - 1.001 (above): where that is not below the floor's distance the ten folds -- forty square roots -- are left out
		const vec3 v = p - V3(0.f, 1.f, 0.f);
		const float k = max1(d, 0.f) + 1.02f;
		if (dot(v, v) >= k * k) return d;
		return min1(d, tetra(v));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		if (on_surface(U, tetra(sp.pos - V3(0.f, 1.f, 0.f))))
		{
			m.diffuse.x = 0.9f;
			m.diffuse.y = 0.7f;
			m.diffuse.z = 0.2f;
			set_rgb(m.specular, 0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// =========================================================================================
struct SceneNeon
{
	static const char *name() { return "neon"; }
	static constexpr int tile_w_log2 = 4; // 16 x 4 pixels per wave: 1.335 -> 1.290 ms at 4K (sdfr_render_pixel.h)
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables()
	{
		return "VAR_r1(min = 0.2, max = 2, start = 1) VAR_r2(min = 0.005, max = 0.1, start = 0.01) VAR_spacing(min = 0.01, max = 0.2, start = 0.1) "
			   "VAR_red(min = 0, max = 3, start = 0.1, step = 0.05) VAR_green(min = 0, max = 3, start = 1.0, step = 0.05) "
			   "VAR_blue(min = 0, max = 3, start = 0.2, step = 0.05)";
	}
	enum { SU_MIRROR_S = 0, SU_MIRROR_C = 1 };
	static SDF_HD void prepare(FrameU &U)
	{
		const vec2 sc = sincos1(0.3f); // the mirror's fixed yaw
		U.su[SU_MIRROR_S] = sc.x;
		U.su[SU_MIRROR_C] = sc.y;
	}
	struct RayInv { GroundInv ground; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		return r;
	}
	// thin rings at quantised latitudes of a sphere
	static SDF_HD float ring_sphere(vec3 p, float spacing, float r1, float r2)
	{
		vec3 hp = normalize(p) * r1;
		const float y = hp.y;
		const float x = length(V2(hp.x, hp.z));
		float angle = atan21(y, x);
		angle = rne1(angle / spacing) * spacing;
		const vec2 sc = sincos1(angle);
		hp.y = sc.x / sc.y * x; // tan(angle) * x
		hp = normalize(hp) * r1;
		return length(p - hp) - r2;
	}
	struct Objects { float rings, mirror, border; };
	static SDF_HD Objects eval_objects(const FrameU &U, vec3 p)
	{
		Objects o;
		o.rings = ring_sphere(p - V3(0.f, 2.f, 0.f), U.scene_var[2], U.scene_var[0], U.scene_var[1]);
		vec3 mp = p - V3(0.f, 2.f, 2.75f);
		const vec2 r = rot2(V2(mp.x, mp.z), U.su[SU_MIRROR_S], U.su[SU_MIRROR_C]);
		mp = V3(r.x, mp.y, r.y);
		o.mirror = sd_box(mp, V3(1.f, 1.7f, 0.05f));
		o.border = sd_box(mp, V3(1.05f, 1.75f, 0.04f));
		return o;
	}
	// The rings are tubes of radius r2 about circles ON the sphere of radius r1 about (0, 2, 0): ring_sphere() is the distance to a point
	// hp of that sphere minus r2, and |p - hp| >= | |p - c| - r1 | (triangle), so rings >= | |p - c| - r1 | - r2 outside and inside the
	// sphere alike.  Where that (0.01 of slack for the rounding of hp) is not below floor, pane and frame, the rings -- an atan2, a sincos, two
	// normalisations, a division -- cannot lower the minimum and are left out.  Checked numerically in tests/test_scene_bounds_cpu.py.
	static SDF_HD float rings_lower_bound(const FrameU &U, vec3 p) { return abs1(length(p - V3(0.f, 2.f, 0.f)) - U.scene_var[0]) - U.scene_var[1] - 0.01f; }
	// floor + rings (in the ball of radius r1 + r2 about (0, 2, 0)) + pane and frame (half size (1.05, 1.75, 0.05) about (0, 2, 2.75), whatever
	// their yaw: in the ball of radius 2.05 about that point): a ray that does not descend and has both balls behind it or passes them
	// at a distance, or is above both, has nothing left to hit
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &, vec3 p, vec3 dir)
	{
		if (!(dir.y >= 0.f) || !(p.y > 1e-20f)) return false; // (as ray_leaves_floor_and_ball: sdfr_lib.h)
		const float reach = abs1(U.scene_var[0]) + abs1(U.scene_var[1]) + 0.02f;
		if (p.y > 2.f + max1(reach, 1.77f)) return true;
		return ray_passes_ball(p, dir, V3(0.f, 2.f, 0.f), reach) && ray_passes_ball(p, dir, V3(0.f, 2.f, 2.75f), 2.07f);
	}
	static constexpr bool inline_escaped_shadows = true; // shadow rays that escape where they start are not queued (sdfr_pixel.h)
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		vec3 mp = p - V3(0.f, 2.f, 2.75f);
		const vec2 r = rot2(V2(mp.x, mp.z), U.su[SU_MIRROR_S], U.su[SU_MIRROR_C]);
		mp = V3(r.x, mp.y, r.y);
		const float mirror = sd_box(mp, V3(1.f, 1.7f, 0.05f)), border = sd_box(mp, V3(1.05f, 1.75f, 0.04f));
		// min() over the same four values as map(), the rings in their place when they are needed
		if (!(rings_lower_bound(U, p) >= min1(min1(d, mirror), border)))
			d = min1(d, ring_sphere(p - V3(0.f, 2.f, 0.f), U.scene_var[2], U.scene_var[0], U.scene_var[1]));
		d = min1(d, mirror);
		return min1(d, border);
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		const Objects o = eval_objects(U, sp.pos);
		if (on_surface(U, o.rings))
		{
			const vec3 c = V3(U.scene_var[3], U.scene_var[4], U.scene_var[5]);
			m.emissive = c;
			const vec3 h = c / 2.f;
			m.diffuse.x = h.x;
			m.diffuse.y = h.y;
			m.diffuse.z = h.z;
			set_rgb(m.specular, 0.5f);
		}
		else if (on_surface(U, o.mirror))
		{
			m.reflection = V3s(0.8f);
			set_rgb(m.specular, 0.1f);
		}
		else if (on_surface(U, o.border))
		{
			m.diffuse.x = 0.5f;
			m.diffuse.y = 0.5f;
			m.diffuse.z = 0.5f;
			set_rgb(m.specular, 0.5f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

} // namespace sdfr
