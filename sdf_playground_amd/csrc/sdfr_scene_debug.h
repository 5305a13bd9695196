// sdfr_scene_debug.h -- a diagnostic scene of the library's own: four objects that wear the driver's
// debug materials (pshader_sdf.hlsl:430-455: MATERIAL_ITER, MATERIAL_PLAIN, MATERIAL_NORMAL1,
// MATERIAL_NORMAL2; iter_count_to_color, sdf_materials.hlsl:143-186).  In the reference a scene
// author selects them by hand while debugging (README.md:98-104) and no shipped scene does, so the
// reference's scene list (sdfr_scene_count / sdfr_scene_name) does not contain this one: it is
// loaded by name, "debug_materials".  tests/ compile the same text at run time as well.
#pragma once
#include "sdfr_frame.h"
#include "sdfr_lib.h"

namespace sdfr {

struct SceneDebugMaterials
{
	static const char *name() { return "debug_materials"; }
	static const char *variables() { return ""; }
	static SDF_HD void prepare(FrameU &) {}
	struct RayInv { GroundInv ground; };
	static SDF_HD RayInv ray_setup(const FrameU &U, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		return r;
	}
	// exact primitives on purpose: grazing rays take many steps, so the iteration colours vary
	static SDF_HD float ball(vec3 p) { return sd_sphere(p - V3(-1.8f, 0.6f, 0.f), 0.6f); }
	static SDF_HD float block(vec3 p) { return sd_box(p - V3(-0.6f, 0.5f, 0.f), V3(0.4f, 0.5f, 0.4f)); }
	static SDF_HD float ring(vec3 p) { return sd_torus_xy(p - V3(0.6f, 0.7f, 0.f), 0.45f, 0.2f); }
	static SDF_HD float drum(vec3 p) { return sd_capped_cylinder(p - V3(1.8f, 0.6f, 0.f), 0.6f, 0.4f) - 0.05f; }
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		d = min1(d, ball(p));
		d = min1(d, block(p));
		d = min1(d, ring(p));
		return min1(d, drum(p));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		if (on_surface(U, ball(sp.pos)))
		{
			m.id = MAT_ITER;
		}
		else if (on_surface(U, block(sp.pos)))
		{
			// unlit plain colour; the mirror coat makes secondary rays reach the other debug materials
			m.id = MAT_PLAIN;
			m.diffuse = V4(0.2f, 0.6f, 0.9f, 1.f);
			m.reflection = V3s(0.3f);
		}
		else if (on_surface(U, ring(sp.pos)))
		{
			m.id = MAT_NORMAL1;
		}
		else if (on_surface(U, drum(sp.pos)))
		{
			// a material normal blended in by a quarter: n = lerp(geometric normal, m.normal.xyz, m.normal.w)
			m.id = MAT_NORMAL2;
			m.normal = V4(0.f, 1.f, 0.f, 0.25f);
		}
	}
	static SDF_HD bool light(const FrameU &U, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

// normal_test: a diagnostic scene for the one callback of the scene ABI that no reference scene fills in, map_normal
// (sdf_structs.hlsl:39-52, pshader_sdf.hlsl:318-330, :520) -- here Scene::normal (SceneNormal, sdfr_pixel.h).  A ball
// whose normal is analytic (VAR_analytic), a mirror-coated block and a drum whose normals are sampled VAR_round apart
// ("larger than usual values lead to rounded corners"), a block left at the default.  Loaded by name, like debug_materials;
// oracle twin: oracle/test_scenes.h.
struct SceneNormalTest
{
	static const char *name() { return "normal_test"; }
	static const char *variables() { return "VAR_round(min = 0.0001, max = 0.05, start = 0.01) VAR_analytic(min = 0, max = 1, step = 1, start = 1)"; }
	enum { SV_ROUND = 0, SV_ANALYTIC = 1 };
	static SDF_HD void prepare(FrameU &) {}
	struct RayInv { GroundInv ground; };
	static SDF_HD RayInv ray_setup(const FrameU &, vec3 dir, const RayFlags &)
	{
		RayInv r;
		r.ground = ground_setup(dir);
		return r;
	}
	static SDF_HD float ball(vec3 p) { return sd_sphere(p - V3(-1.6f, 0.7f, 0.2f), 0.7f); }
	static SDF_HD float block(vec3 p) { return sd_box(p - V3(0.f, 0.5f, 0.f), V3(0.5f, 0.5f, 0.5f)); }
	static SDF_HD float drum(vec3 p) { return sd_capped_cylinder(p - V3(1.5f, 0.45f, -0.3f), 0.45f, 0.4f); }
	static SDF_HD float plain_block(vec3 p) { return sd_box(p - V3(0.4f, 0.3f, -1.6f), V3(0.3f, 0.3f, 0.3f)); }
	static SDF_HD float dist(const FrameU &, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		d = min1(d, ball(p));
		d = min1(d, block(p));
		d = min1(d, drum(p));
		return min1(d, plain_block(p));
	}
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		ground_material(U, sp, m);
		if (on_surface(U, ball(sp.pos)))
		{
			m.diffuse = V4(0.8f, 0.3f, 0.2f, 1.f);
			set_rgb(m.specular, 1.f);
			m.specular.w = 20.f;
		}
		else if (on_surface(U, block(sp.pos)))
		{
			m.diffuse = V4(0.2f, 0.3f, 0.8f, 1.f);
			set_rgb(m.specular, 0.5f);
			m.reflection = V3s(0.4f);
		}
		else if (on_surface(U, drum(sp.pos)))
		{
			m.id = MAT_NORMAL2; // the geometric normal as colour: the rounded rim shows
		}
		else if (on_surface(U, plain_block(sp.pos)))
		{
			m.diffuse = V4(0.3f, 0.8f, 0.3f, 1.f);
			set_rgb(m.specular, 0.5f);
		}
	}
	// map_normal: which object the hit point lies on, with a tolerance well above the march's dist_eps
	static SDF_HD void normal(const FrameU &U, const SurfacePoint &sp, NormalOut &no)
	{
		if (U.scene_var[SV_ANALYTIC] != 0.f && abs1(ball(sp.pos)) < 0.01f)
		{
			no.use_normal = true;
			no.normal = normalize(sp.pos - V3(-1.6f, 0.7f, 0.2f));
		}
		else if (abs1(block(sp.pos)) < 0.01f || abs1(drum(sp.pos)) < 0.01f)
		{
			no.sample_dist = U.scene_var[SV_ROUND];
		}
	}
	static SDF_HD bool light(const FrameU &, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

} // namespace sdfr
