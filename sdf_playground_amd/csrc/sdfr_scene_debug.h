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
	static SDF_HD RayInv ray_setup(const FrameU &, vec3 dir, const RayFlags &)
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
	static SDF_HD float dist(const FrameU &, const RayInv &R, vec3 p, vec3, bool fast)
	{
		float d = min1(3e38f, ground_dist(p, fast, R.ground));
		d = min1(d, ball(p));
		d = min1(d, block(p));
		d = min1(d, ring(p));
		return min1(d, drum(p));
	}
	static SDF_HD void material(const FrameU &, const SurfacePoint &sp, Material &m)
	{
		ground_material(sp, m);
		if (on_surface(ball(sp.pos)))
		{
			m.id = MAT_ITER;
		}
		else if (on_surface(block(sp.pos)))
		{
			// unlit plain colour; the mirror coat makes secondary rays reach the other debug materials
			m.id = MAT_PLAIN;
			m.diffuse = V4(0.2f, 0.6f, 0.9f, 1.f);
			m.reflection = V3s(0.3f);
		}
		else if (on_surface(ring(sp.pos)))
		{
			m.id = MAT_NORMAL1;
		}
		else if (on_surface(drum(sp.pos)))
		{
			// a material normal blended in by a quarter: n = lerp(geometric normal, m.normal.xyz, m.normal.w)
			m.id = MAT_NORMAL2;
			m.normal = V4(0.f, 1.f, 0.f, 0.25f);
		}
	}
	static SDF_HD bool light(const FrameU &, int i, Light &L) { return sun_light(i, L); }
	static SDF_HD float ambient() { return 0.075f; }
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t) { return sky_color(dir, U.sky_s, U.sky_c); }
};

} // namespace sdfr
