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
	static constexpr int waves_per_simd = 5; // ~1900 instructions per evaluation: registers over residency (sdfr_pixel_kernel.h); 16.9 -> 15.7 ms at 4K
	static constexpr bool persistent_tiles = true; // with waves that retire: 14.75 -> 14.25 ms at 4K
	static constexpr int retire_after = 4;
	static const char *variables() { return ""; }
	enum { SU_DRIFT = 0 };
	static SDF_HD void prepare(FrameU &U) { U.su[SU_DRIFT] = U.stime / 10.f * 0.4f; }

	struct RayInv { GroundInv ground; vec2 dir2; bool rising; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		r.dir2 = normalize(V2(dir.x, dir.z));
		r.rising = dir.y >= 0.f;
		return r;
	}
	// the scene itself looks at its trees only below the canopy's bounding plane (dist: bounding < 0.1, i.e. y < 2.1);
	// above it the distance is min(that plane, floor), both behind a ray that does not descend
	static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &R, vec3 p, vec3) { return R.rising && p.y > 2.11f; }

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
	// voronoi lattice, first half: the nine candidate sites around uv, the nearest (id, vector to it) and the
	// distances to the nearest and the second nearest
	struct Lattice { vec2 sites[9]; vec2 local, best_site; float best, second; };
	static SDF_HD void lattice_sites(vec2 uv, float max_offset, Lattice &L, vec2 *id, vec2 *to_site)
	{
		const vec2 cell = floor(uv);
		L.local = (uv - cell) - 0.5f;
		L.best = 10.f;
		L.second = 10.f;
		L.best_site = V2(0.f, 0.f);
		// the reference hashes the nine neighbour sites twice (once per loop); same inputs, same
		// values: they are kept from the first pass
		bool shared = false;
#if defined(__HIP_DEVICE_COMPILE__)
		// mostly every lane of the wave stands in the same lattice cell: then nine lanes hash one site each (two PCG
		// hashes: a sixth of an evaluation, done nine times over otherwise) and pass them round (WaveShare, sdfr_lib.h)
		{
			uint32_t rank;
			if (WaveShare::agree(cell.x, cell.y, 0.f, 9u, &rank))
			{
				float *slots = WaveShare::slots();
				if (rank < 9u)
				{
					const uint32_t col = rank / 3u;
					const vec2 off = V2((float)col - 1.f, (float)(rank - 3u * col) - 1.f);
					const vec2 site = off + voronoi_site(cell + off) * max_offset;
					slots[2u * rank] = site.x;
					slots[2u * rank + 1u] = site.y;
				}
				WaveShare::publish();
#pragma unroll
				for (int k = 0; k < 9; ++k) L.sites[k] = V2(slots[2 * k], slots[2 * k + 1]);
				WaveShare::release();
				shared = true;
			}
		}
#endif
		if (!shared)
		{
#pragma unroll
			for (int x = -1; x < 2; ++x)
#pragma unroll
				for (int y = -1; y < 2; ++y)
				{
					const vec2 off = V2((float)x, (float)y);
					L.sites[(x + 1) * 3 + (y + 1)] = off + voronoi_site(cell + off) * max_offset;
				}
		}
		// the nearest site: only the site and its cell offset are selected per candidate (four selects instead of seven);
		// id and the vector to the site are formed from them afterwards, by the expressions the reference's loop uses
		vec2 best_off = V2(0.f, 0.f);
#pragma unroll
		for (int x = -1; x < 2; ++x)
#pragma unroll
			for (int y = -1; y < 2; ++y)
			{
				const vec2 site = L.sites[(x + 1) * 3 + (y + 1)];
				const float len = length(site - L.local);
				L.second = min1(L.second, max1(len, L.best)); // not part of the reference: feeds border_lower_bound
				if (len < L.best) { L.best = len; L.best_site = site; best_off = V2((float)x, (float)y); }
			}
		*id = cell + best_off;
		*to_site = L.best_site - L.local;
	}
	// second half: the distance to the border of the nearest site's cell along dir
	static SDF_HD float lattice_border(const Lattice &L, vec2 dir)
	{
		float edge = 10.f;
#pragma unroll
		for (int k = 0; k < 9; ++k)
		{
			const vec2 mid = (L.sites[k] + L.best_site) * 0.5f;
			const vec2 n = normalize(L.best_site - mid);
			const float e = abs1(dot(n, L.local - mid));
			edge = min1(edge, e / max1(dot(n, -dir), 0.0001f));
		}
		return edge;
	}
	// A lower bound of lattice_border() from the first half alone.  Candidate k contributes e_k / den_k with
	// e_k the distance from `local` to the bisector of (nearest site, site k) and den_k <= |n||dir| <= 1 + 4e-7.
	// With a = |local - nearest|, b = |local - site k| and D = |nearest - site k| <= a + b (triangle), the
	// bisector is (b^2 - a^2) / 2D >= (b - a) / 2 away, and b >= the second smallest length.  The sites differ
	// by >= 0.4 (unit lattice, offsets <= 0.3), all operands are <= 2, so rounding moves the computed e_k, the
	// lengths and den_k by a few 1e-7: (second - best) / 2, shrunk by 1e-5 relative and 2e-5 absolute, stays
	// below every computed quotient; the candidate that is the nearest site itself gives 0 / 0 = NaN, which
	// min() drops, and the initial 10 is above any bound (lengths < 2.2).  NaN in, NaN out: the caller then
	// evaluates the border.  Checked numerically in tests/test_scene_bounds_cpu.py.
	static SDF_HD float border_lower_bound(const Lattice &L) { return max1((L.second - L.best) * 0.499995f - 2e-5f, 0.f); }
	// slide for 10 s, then hop for 1 s: (progress along the slide, hop height).  The reference's hop(10, 1, t)
	// with its divisions by the constants 11 and 10 as div_c (exact: tests/test_gpu_math.py, SCENE_DIVISORS)
	static SDF_HD vec2 hop(float t)
	{
		const float slide_time = 10.f, jump_time = 1.f;
		const float total = slide_time + jump_time;
		const float cycle = t - floor1(div_c(t, total, 1.0f / total)) * total;
		if (cycle < slide_time) return V2(div_c(cycle, slide_time, 1.0f / slide_time), 0.f);
		const float j = (cycle - slide_time) / jump_time;
		return V2(1.f - j, 4.f * (j - j * j));
	}

	struct Objects { float bounding, tree, leaves, eye, pupil, noise; };
	// `others`: what dist() takes the minimum with besides the objects (the ground), or a negative number when
	// the caller does not look at distances above U.dist_eps at all (material()).  The reference clamps tree
	// and leaves to guard = border * spacing + 0.1 so that no step leaves the lattice cell; border >= 0, so
	// guard >= 0.1, and guard is monotone in border.  Where the guard's lower bound is not below the minimum of
	// everything else, the clamp cannot change the scene distance (nor any on_surface(U, ) test, 0.1 > U.dist_eps)
	// and the second half of the lattice -- nine normalisations and divisions, a quarter of an evaluation -- is
	// left out: near a tree, i.e. for the steps that close in on a hit, its six gradient probes and material().
	static SDF_HD Objects eval_objects(const FrameU &U, vec3 p, vec2 dir2, float bounding, float others)
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
			Lattice L;
			lattice_sites(V2(div_c(pos.x, spacing, 1.0f / spacing), div_c(pos.z, spacing, 1.0f / spacing)), 0.3f, L, &id, &cell_pos);
			o.noise = sin1(id.x * 356.12f + id.y + 82.6f) * 0.5f + 0.5f;

			const vec2 hopped = hop(U.stime + o.noise * 10.f);
			const vec2 jump = V2(div_c(hopped.x, spacing, 1.0f / spacing), div_c(hopped.y, spacing, 1.0f / spacing));
			cell_pos.y = cell_pos.y - (jump.x * 0.4f - 0.05f);
			const vec2 cs = cell_pos * spacing;
			vec3 tree_pos = V3(cs.x, pos.y - jump.y, cs.y);
			const vec2 cr = op_rotate(cell_pos, o.noise) * spacing;
			tree_sdf(V3(cr.x, pos.y - jump.y, cr.y), o.noise, &o.tree, &o.leaves);
			tree_pos.x = abs1(tree_pos.x);
			o.eye = sd_sphere(tree_pos - V3(0.2f, 1.f, -0.5f), 0.12f);
			o.pupil = sd_sphere(tree_pos - V3(0.2f, 1.f, -0.59f), 0.05f);
			// never step past the border of the current lattice cell
			const float nearest = min1(min1(min1(o.tree, o.leaves), min1(o.eye, o.pupil)), others);
			if (!(border_lower_bound(L) * spacing + 0.1f >= nearest))
			{
				const float guard = lattice_border(L, dir2) * spacing + 0.1f;
				o.tree = min1(o.tree, guard);
				o.leaves = min1(o.leaves, guard);
			}
		}
		return o;
	}
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		const float bounding = ground_dist(p - V3(0.f, 2.f, 0.f), fast, R.ground);
		const float ground = ground_dist(p, fast, R.ground);
		const Objects o = eval_objects(U, p, R.dir2, bounding, ground);
		float d = 3e38f;
		if (bounding >= 0.1f) d = min1(d, bounding);
		d = min1(d, o.tree);
		d = min1(d, o.leaves);
		d = min1(d, o.eye);
		d = min1(d, o.pupil);
		return min1(d, ground);
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		const float bounding = dot(sp.pos - V3(0.f, 2.f, 0.f), V3(0.f, 1.f, 0.f));
		const Objects o = eval_objects(U, sp.pos, V2(0.f, 0.f), bounding, -1.f);
		if (on_surface(U, o.tree))
		{
			m.diffuse.x = 0.5f;
			m.diffuse.y = 0.25f;
			m.diffuse.z = 0.1f;
			set_rgb(m.specular, 0.15f);
		}
		else if (on_surface(U, o.leaves))
		{
			const vec3 green = lerp(V3(0.2f, 0.9f, 0.2f), V3(0.3f, 0.5f, 0.2f), o.noise);
			m.diffuse.x = green.x;
			m.diffuse.y = green.y;
			m.diffuse.z = green.z;
			set_rgb(m.specular, 0.15f);
		}
		else if (on_surface(U, o.eye))
		{
			set_rgb(m.diffuse, 0.9f);
			set_rgb(m.specular, 0.15f);
		}
		else if (on_surface(U, o.pupil))
		{
			set_rgb(m.diffuse, 0.1f);
			set_rgb(m.specular, 0.15f);
		}
		else if (on_surface(U, dot(sp.pos, V3(0.f, 1.f, 0.f))))
		{
			const float turb = turbulence3(sp.pos);
			const vec3 soil = lerp(V3(218.f, 173.f, 136.f) / 255.f, V3(140.f, 90.f, 60.f) / 255.f, turb) * 0.6f;
			m.diffuse.x = soil.x;
			m.diffuse.y = soil.y;
			m.diffuse.z = soil.z;
			set_rgb(m.specular, 0.05f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L)
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
