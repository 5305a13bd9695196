// oracle/test_scenes.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Scenes that exist for tests only; none of them restates a reference scene.
//
// debug_materials: the driver's debug materials MATERIAL_ITER / MATERIAL_PLAIN / MATERIAL_NORMAL1 /
// MATERIAL_NORMAL2 (pshader_sdf.hlsl:430-455, sdf_materials.hlsl:143-186) are selected by a scene's
// material_id, but no scene of the reference emits them (its authors switch them on by hand while
// debugging, README.md:98-104).  This scene puts one of each on four objects over the usual
// floor, in the shape a reference scene would have (map_groundplane, OBJECT / MATERIAL chain), so the
// four branches of ps_main are driven by the same arithmetic on the oracle and on the kernels.
#pragma once
#include "scenes.h"

namespace orc {

struct SceneDebugMaterials
{
	static const char *name() { return "debug_materials"; }
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		// exact primitives on purpose: grazing rays take many steps, so the iteration colours vary
		real ball = sdSphere(geometry.pos - float3(real(-1.8f), real(0.6f), real(0.f)), real(0.6f));
		real block = sdBox(geometry.pos - float3(real(-0.6f), real(0.5f), real(0.f)), float3(real(0.4f), real(0.5f), real(0.4f)));
		real ring = sdTorusXY(geometry.pos - float3(real(0.6f), real(0.7f), real(0.f)), real(0.45f), real(0.2f));
		real drum = sdCappedCylinder(geometry.pos - float3(real(1.8f), real(0.6f), real(0.f)), real(0.6f), real(0.4f)) - real(0.05f);
		if (geometry_step)
		{
			object_add(output_scene_distance, ball);
			object_add(output_scene_distance, block);
			object_add(output_scene_distance, ring);
			object_add(output_scene_distance, drum);
		}
		else if (material_hit(ball))
		{
			material_output.material_id = MATERIAL_ITER;
		}
		else if (material_hit(block))
		{
			// unlit plain colour; the mirror coat makes secondary rays reach the other debug materials
			material_output.material_id = MATERIAL_PLAIN;
			material_output.diffuse_color = float4(real(0.2f), real(0.6f), real(0.9f), real(1.f));
			material_output.reflection_color = float3(real(0.3f));
		}
		else if (material_hit(ring))
		{
			material_output.material_id = MATERIAL_NORMAL1;
		}
		else if (material_hit(drum))
		{
			// a material normal blended in by a quarter: new_normal = lerp(obj_normal, normal.xyz, normal.w)
			material_output.material_id = MATERIAL_NORMAL2;
			material_output.normal = float4(real(0.f), real(1.f), real(0.f), real(0.25f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// normal_test: the one scene callback no reference scene fills in -- map_normal (sdf_structs.hlsl:39-52,
// pshader_sdf.hlsl:318-330: a scene may hand the driver the normal itself, and / or widen the spacing of the
// forward-difference samples, "larger than usual values lead to rounded corners"; the same spacing then moves the
// start of the shadow rays, pshader_sdf.hlsl:520).  In the shape a reference scene would have: a ball whose normal
// is analytic (VAR_analytic), a mirror-coated block and a drum whose normals are sampled VAR_round apart, a second
// block left at the default.
struct SceneNormalTest
{
	static const char *name() { return "normal_test"; }
	static real ball(float3 p) { return sdSphere(p - float3(real(-1.6f), real(0.7f), real(0.2f)), real(0.7f)); }
	static real block(float3 p) { return sdBox(p - float3(real(0.f), real(0.5f), real(0.f)), float3(real(0.5f), real(0.5f), real(0.5f))); }
	static real drum(float3 p) { return sdCappedCylinder(p - float3(real(1.5f), real(0.45f), real(-0.3f)), real(0.45f), real(0.4f)); }
	static real plain_block(float3 p) { return sdBox(p - float3(real(0.4f), real(0.3f), real(-1.6f)), float3(real(0.3f), real(0.3f), real(0.3f))); }
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		real d_ball = ball(geometry.pos), d_block = block(geometry.pos), d_drum = drum(geometry.pos), d_plain = plain_block(geometry.pos);
		if (geometry_step)
		{
			object_add(output_scene_distance, d_ball);
			object_add(output_scene_distance, d_block);
			object_add(output_scene_distance, d_drum);
			object_add(output_scene_distance, d_plain);
		}
		else if (material_hit(d_ball))
		{
			material_output.diffuse_color = float4(real(0.8f), real(0.3f), real(0.2f), real(1.f));
			set_rgb(material_output.specular_color, real(1.f));
			material_output.specular_color.w = real(20.f);
		}
		else if (material_hit(d_block))
		{
			material_output.diffuse_color = float4(real(0.2f), real(0.3f), real(0.8f), real(1.f));
			set_rgb(material_output.specular_color, real(0.5f));
			material_output.reflection_color = float3(real(0.4f));
		}
		else if (material_hit(d_drum))
		{
			// the geometric normal as colour: the rounded rim shows
			material_output.material_id = MATERIAL_NORMAL2;
		}
		else if (material_hit(d_plain))
		{
			material_output.diffuse_color = float4(real(0.3f), real(0.8f), real(0.3f), real(1.f));
			set_rgb(material_output.specular_color, real(0.5f));
		}
	}
	static void map_normal(const Frame &F, const GeometryInput &geometry, NormalOutput &normal_output)
	{
		// which object the hit point lies on, with a tolerance well above the march's dist_eps
		if (F.scene_var[1] != real(0.f) && r_abs(ball(geometry.pos)) < real(0.01f))
		{
			normal_output.use_normal = true;
			normal_output.normal = normalize(geometry.pos - float3(real(-1.6f), real(0.7f), real(0.2f)));
			// ... and, like any reference scene would, leaves the spacing alone: the shadow rays start max(shadow_eps, grad_eps) off
		}
		else if (r_abs(block(geometry.pos)) < real(0.01f) || r_abs(drum(geometry.pos)) < real(0.01f))
		{
			normal_output.normal_sample_dist = F.scene_var[0]; // VAR_round
		}
	}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

} // namespace orc
