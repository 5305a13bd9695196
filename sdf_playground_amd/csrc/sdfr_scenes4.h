// sdfr_scenes4.h -- the `tree` scene functor (Engine/shader/scenes/sdf_scene_tree.hlsl): a
// forest of hopping, googly-eyed trees placed on a voronoi lattice (device code,
// host-compilable).
#pragma once
#include "sdfr_scenes3.h"

namespace sdfr {

struct SceneTree
{
	static const char *name() { return "tree"; }
	static constexpr bool shadow_hits_need_normal = false; // material() does not read sp.normal
	static const char *variables() { return ""; }
	enum { SU_DRIFT = 0 };
	static SDF_HD void prepare(FrameU &U) { U.su[SU_DRIFT] = U.stime / 10.f * 0.4f; }

	struct RayInv { GroundInv ground; vec2 dir2; };
	static SDF_HD RayInv ray_setup(const FrameU &, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		r.dir2 = normalize(V2(dir.x, dir.z));
		return r;
	}

	// a truncated cone standing on the origin
	static SDF_HD float branch(vec3 p, float h2, float r1, float r2)
	{
		const vec2 p2 = V2(length(V2(p.x, p.z)), p.y);
		const float side = dot(p2 - V2(r1, 0.f), normalize(V2(h2, r1 - r2)));
		return max1(max1(side, -p.y), p.y - h2);
	}
	// nine generations of branches, each carrying a leaf ball
	static SDF_HD void tree_sdf(vec3 p, float noise, float *tree, float *leaves)
	{
		float tree_scale = 1.f, leaf_scale = 1.f;
		float t = 3e30f, l = 3e30f;
		const float angle3 = 90.f - noise * 10.f;
		const float height0 = 0.33f + noise * 0.05f;
		const float ball = 0.07f - noise * 0.02f;
		const vec2 twist = sincos1(angle3 / 180.f * SDFR_PI); // the same twist at every generation
#pragma unroll
		for (int i = 0; i < 9; ++i)
		{
			// tree_scale = 1.4^-i is a compile-time constant after unrolling: the division is a div_c
			// (exact for all nine constants: tests/test_gpu_math.py, SCENE_DIVISORS)
			const float inv_scale = 1.0f / tree_scale;
			const vec3 ps = V3(div_c(p.x, tree_scale, inv_scale), div_c(p.y, tree_scale, inv_scale), div_c(p.z, tree_scale, inv_scale));
			t = op_smin_c(t, branch(ps, 1.f, 0.1f, 0.05f) * tree_scale, 0.01f, 1.0f / 0.01f);
			l = min1(l, sd_sphere(ps - V3(0.f, 1.f + ball * leaf_scale, 0.f), ball * leaf_scale) * tree_scale);

			const float height = i == 0 ? height0 : 0.41f;
			p.y = p.y - height * tree_scale;
			p.x = abs1(p.x);
			p.z = abs1(p.z);
			if (p.x > p.z && i == 0) { const float s = p.x; p.x = p.z; p.z = s; }
			p.z = p.z + 0.075f * tree_scale;
			const float angle = i == 0 ? 35.f : 34.f;
			const vec2 yz = op_rotate(V2(p.y, p.z), -angle / 180.f * SDFR_PI);
			p.y = yz.x;
			p.z = yz.y;
			const vec2 xz = rot2(V2(p.x, p.z), twist.x, twist.y);
			p.x = xz.x;
			p.z = xz.y;
			tree_scale = tree_scale / 1.4f;
			leaf_scale = leaf_scale * 1.3f;
		}
		*tree = t;
		*leaves = l;
	}
	// voronoi lattice: nearest site (id, vector to it) and the distance to the cell border along dir
	static SDF_HD void lattice(vec2 uv, vec2 dir, float max_offset, vec2 *id, vec2 *to_site, float *border)
	{
		const vec2 cell = floor(uv);
		const vec2 local = (uv - cell) - 0.5f;
		float best = 10.f, edge = 10.f;
		vec2 best_site = V2(0.f, 0.f);
		// the reference hashes the nine neighbour sites twice (once per loop); same inputs, same
		// values: they are kept from the first pass
		vec2 sites[9];
#pragma unroll
		for (int x = -1; x < 2; ++x)
#pragma unroll
			for (int y = -1; y < 2; ++y)
			{
				const vec2 off = V2((float)x, (float)y);
				const vec2 site = off + voronoi_site(cell + off) * max_offset;
				sites[(x + 1) * 3 + (y + 1)] = site;
				const vec2 v = site - local;
				const float len = length(v);
				if (len < best) { best = len; *id = cell + off; best_site = site; *to_site = v; }
			}
#pragma unroll
		for (int k = 0; k < 9; ++k)
		{
			const vec2 mid = (sites[k] + best_site) * 0.5f;
			const vec2 n = normalize(best_site - mid);
			const float e = abs1(dot(n, local - mid));
			edge = min1(edge, e / max1(dot(n, -dir), 0.0001f));
		}
		*border = edge;
	}
	// slide for `slide_time`, then hop for `jump_time`: (progress along the slide, hop height)
	static SDF_HD vec2 hop(float slide_time, float jump_time, float t)
	{
		const float total = slide_time + jump_time;
		const float cycle = t - floor1(t / total) * total;
		if (cycle < slide_time) return V2(cycle / slide_time, 0.f);
		const float j = (cycle - slide_time) / jump_time;
		return V2(1.f - j, 4.f * (j - j * j));
	}

	struct Objects { float bounding, tree, leaves, eye, pupil, noise; };
	static SDF_HD Objects eval_objects(const FrameU &U, vec3 p, vec2 dir2, float bounding)
	{
		Objects o;
		o.bounding = bounding;
		o.tree = 1e30f;
		o.leaves = 1e30f;
		o.eye = 1e30f;
		o.pupil = 1e30f;
		o.noise = 0.f;
		if (bounding < 0.1f) // only below the canopy's bounding plane
		{
			vec3 pos = p;
			pos.z = pos.z - U.su[SU_DRIFT];
			const float spacing = 2.2f;
			vec2 id = V2(0.f, 0.f), cell_pos = V2(0.f, 0.f);
			float border;
			lattice(V2(pos.x, pos.z) / spacing, dir2, 0.3f, &id, &cell_pos, &border);
			o.noise = sin1(id.x * 356.12f + id.y + 82.6f) * 0.5f + 0.5f;

			const vec2 jump = hop(10.f, 1.f, U.stime + o.noise * 10.f) / spacing;
			cell_pos.y = cell_pos.y - (jump.x * 0.4f - 0.05f);
			const vec2 cs = cell_pos * spacing;
			vec3 tree_pos = V3(cs.x, pos.y - jump.y, cs.y);
			const vec2 cr = op_rotate(cell_pos, o.noise) * spacing;
			tree_sdf(V3(cr.x, pos.y - jump.y, cr.y), o.noise, &o.tree, &o.leaves);
			tree_pos.x = abs1(tree_pos.x);
			o.eye = sd_sphere(tree_pos - V3(0.2f, 1.f, -0.5f), 0.12f);
			o.pupil = sd_sphere(tree_pos - V3(0.2f, 1.f, -0.59f), 0.05f);
			// never step past the border of the current lattice cell
			const float guard = border * spacing + 0.1f;
			o.tree = min1(o.tree, guard);
			o.leaves = min1(o.leaves, guard);
		}
		return o;
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		const float bounding = ground_dist(p - V3(0.f, 2.f, 0.f), fast, R.ground);
		const Objects o = eval_objects(U, p, R.dir2, bounding);
		float d = 3e38f;
		if (bounding >= 0.1f) d = min1(d, bounding);
		d = min1(d, o.tree);
		d = min1(d, o.leaves);
		d = min1(d, o.eye);
		d = min1(d, o.pupil);
		return min1(d, ground_dist(p, fast, R.ground));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		const float bounding = dot(sp.pos - V3(0.f, 2.f, 0.f), V3(0.f, 1.f, 0.f));
		const Objects o = eval_objects(U, sp.pos, normalize(V2(sp.dir.x, sp.dir.z)), bounding);
		if (on_surface(o.tree))
		{
			m.diffuse.x = 0.5f;
			m.diffuse.y = 0.25f;
			m.diffuse.z = 0.1f;
			set_rgb(m.specular, 0.15f);
		}
		else if (on_surface(o.leaves))
		{
			const vec3 green = lerp(V3(0.2f, 0.9f, 0.2f), V3(0.3f, 0.5f, 0.2f), o.noise);
			m.diffuse.x = green.x;
			m.diffuse.y = green.y;
			m.diffuse.z = green.z;
			set_rgb(m.specular, 0.15f);
		}
		else if (on_surface(o.eye))
		{
			set_rgb(m.diffuse, 0.9f);
			set_rgb(m.specular, 0.15f);
		}
		else if (on_surface(o.pupil))
		{
			set_rgb(m.diffuse, 0.1f);
			set_rgb(m.specular, 0.15f);
		}
		else if (on_surface(dot(sp.pos, V3(0.f, 1.f, 0.f))))
		{
			const float turb = turbulence3(sp.pos);
			const vec3 soil = lerp(V3(218.f, 173.f, 136.f) / 255.f, V3(140.f, 90.f, 60.f) / 255.f, turb) * 0.6f;
			m.diffuse.x = soil.x;
			m.diffuse.y = soil.y;
			m.diffuse.z = soil.z;
			set_rgb(m.specular, 0.05f);
		}
	}
	static SDF_HD bool light(const FrameU &, int i, Light &L)
	{
		if (i != 0) return false;
		L.pos = V3(-1.f, -1.f, 1.2f);
		L.directional = true;
		L.color = V3(1.f, 1.f, 1.f) * 1.3f;
		L.extend = 0.f;
		L.falloff = 0.f;
		return true;
	}
	static SDF_HD float ambient() { return 0.2f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

} // namespace sdfr
