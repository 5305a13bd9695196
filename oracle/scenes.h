// oracle/scenes.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates the scene plugins named by BASELINE.json's configs:
//   Engine/shader/scenes/sdf_scene_{fast_sphere,cube_sea,labyrinth,fractal,lense,gems,
//   light_shadows}.hlsl
// Each scene supplies the four callbacks of the scene ABI (pshader_sdf.hlsl:84,
// README.md:114-119): map, map_normal, map_light, map_background.
#pragma once
#include "sdf_lib.h"

namespace orc {

// the one directional light every config scene uses (e.g. sdf_scene_fast_sphere.hlsl:29-34)
inline void default_directional_light(LightOutput *output)
{
	output[0].used = true;
	output[0].pos = float4(real(-1.f), real(-1.f), real(2.f), real(1.f));
	output[0].color = float3(real(1.f), real(1.f), real(1.f));
}

inline void set_rgb(float4 &c, real v) { c.x = v; c.y = v; c.z = v; }

// ---- scenes/sdf_scene_fast_sphere.hlsl ---------------------------------------------
struct SceneFastSphere
{
	static const char *name() { return "fast_sphere"; }
	// :5-23
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		real sphere = sdSphereFast(geometry.pos - float3(real(0.f), real(1.f), real(0.f)), geometry.dir, real(0.5f));
		if (geometry_step)
		{
			object_add(output_scene_distance, sphere);
		}
		else if (material_hit(sphere))
		{
			material_output.diffuse_color = float4(real(0.2f), real(0.7f), real(0.2f), real(1.f));
			set_rgb(material_output.specular_color, real(0.5f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_cube_sea.hlsl ------------------------------------------------
struct SceneCubeSea
{
	static const char *name() { return "cube_sea"; }
	// :5-47
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);

		float3 cell_pos = geometry.pos;
		float2 rep = opRepInf(float2(cell_pos.x, cell_pos.z), float2(real(2.f)));
		cell_pos.x = rep.x;
		cell_pos.z = rep.y;
		float3 cube_pos = cell_pos;
		float2 cell_index = (float2(geometry.pos.x, geometry.pos.z) - float2(cell_pos.x, cell_pos.z)) / real(2.f);
		float2 sometimes_pos = v_round(v_frac(cell_index * real(0.5f) + real(0.25f)));
		bool is_other = sometimes_pos.x < real(0.5f) && sometimes_pos.y < real(0.5f);
		real phase = cell_index.x + cell_index.y * real(0.3f) + F.stime;
		real h = r_sin(phase);
		float2 rot = opRotate(float2(cube_pos.x, cube_pos.z), r_cos(phase) * real(0.4f));
		cube_pos.x = rot.x;
		cube_pos.z = rot.y;

		real cube = sdBox(cube_pos - float3(real(0.f), real(2.f) + h, real(0.f)), is_other ? real(0.25f) : real(0.5f)) - real(0.15f);
		real guard = sdLimit2(float2(cell_pos.x, cell_pos.z), float2(geometry.dir.x, geometry.dir.z), float2(real(2.01f)));

		if (geometry_step)
		{
			object_add(output_scene_distance, cube);
			object_add(output_scene_distance, guard);
		}
		else if (material_hit(cube))
		{
			if (is_other)
			{
				material_output.diffuse_color = float4(real(0.8f), real(0.2f), real(0.2f), real(1.f));
				set_rgb(material_output.specular_color, real(1.f));
			}
			else
			{
				material_output.diffuse_color = float4(real(0.6f), real(0.5f), real(0.2f), real(1.f));
				set_rgb(material_output.specular_color, real(1.f));
				material_output.reflection_color = float3(real(0.25f));
			}
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_labyrinth.hlsl -----------------------------------------------
struct SceneLabyrinth
{
	static const char *name() { return "labyrinth"; }
	// :5-19
	static real vase(float3 pos)
	{
		real obj1 = sdSphere(pos - float3(real(0.f), real(1.89f), real(0.f)), real(0.5f));
		real obj2 = sdCappedCylinder(pos - float3(real(0.f), real(0.8f), real(0.f)), real(0.75f), real(0.2f));
		real obj3 = sdBox(pos - float3(real(0.f), real(0.075f), real(0.f)), float3(real(0.4f), real(0.075f), real(0.4f)));
		real cut_plane1 = sdPlane(pos - float3(real(0.f), real(1.9f), real(0.f)), float3(real(0.f), real(1.f), real(0.f)));
		real cut_plane2 = sdPlane(pos - float3(real(0.f), real(1.5f), real(0.f)), float3(real(0.f), real(-1.f), real(0.f)));

		real d = r_max(obj1, cut_plane2);
		d = opPipeMerge(d, obj2, real(0.1f), real(4.f));
		d = opPipeMerge(d, obj3, real(0.1f), real(4.f));
		d = r_max(d, cut_plane1);
		d = r_max(d, -obj1 - real(0.06f));
		return d;
	}
	// :21-87
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);

		// wall (:26-36)
		float3 wall_pos = geometry.pos;
		float2 rep = opRepInf(float2(wall_pos.x, wall_pos.z), float2(real(20.f), real(20.f)));
		wall_pos.x = r_abs(rep.x);
		wall_pos.z = r_abs(rep.y);
		if (wall_pos.z > wall_pos.x)
		{
			real tmp = wall_pos.x;
			wall_pos.x = wall_pos.z;
			wall_pos.z = tmp;
		}
		real wall1 = sdBox(wall_pos - float3(real(3.5f), real(2.f), real(3.f)), float3(real(1.5f), real(2.f), real(1.f)));
		real wall2 = sdBox(wall_pos - float3(real(7.f), real(2.f), real(5.f)), float3(real(3.f), real(2.f), real(1.f)));
		real wall = r_min(wall1, wall2);

		// vase (:38-42)
		float3 obj_pos = wall_pos;
		obj_pos.x -= real(8.f);
		obj_pos.x = r_abs(obj_pos.x);
		real obj1 = vase(obj_pos - float3(real(1.f), real(0.f), real(3.f)));

		// torch (:44-55); torch_angle/torch_c/torch_s are compile-time constants in the HLSL (a-T.8)
		const real torch_angle = real(15.f) * real(pi) / real(180.f);
		const real torch_c = r_cos(torch_angle), torch_s = r_sin(torch_angle);

		float3 torch_pos = wall_pos - float3(real(5.f), real(2.f), real(3.f));
		float3 wood_pos = torch_pos;
		torch_pos.x -= real(0.3f);
		{
			real wx = wood_pos.x * torch_c - wood_pos.y * torch_s;
			real wy = wood_pos.x * torch_s + wood_pos.y * torch_c;
			wood_pos.x = wx;
			wood_pos.y = wy;
		}
		real wood = sdBox(wood_pos - float3(real(0.f), real(0.6f), real(0.f)), float3(real(0.05f), real(0.5f), real(0.05f)));
		real fire = sdRoundCone(torch_pos, float3(real(0.f), real(1.1f), real(0.f)), float3(real(0.f), real(1.6f), real(0.f)), real(0.15f), real(0.1f));
		real transparent_fire = sdRoundCone(march.last_transparent_pos, float3(real(0.f), real(1.1f), real(0.f)), float3(real(0.f), real(1.6f), real(0.f)), real(0.15f), real(0.1f));

		if (geometry_step)
		{
			object_add(output_scene_distance, wall);
			object_add(output_scene_distance, obj1);
			object_add(output_scene_distance, wood);
			object_add_transparent(output_scene_distance, march, fire, transparent_fire);
		}
		else
		{
			if (material_hit(wall))
			{
				material_output.material_position.x = geometry.pos.x;
				material_output.material_position.y = geometry.pos.y;
				material_output.material_position.z = geometry.pos.z;
				material_output.material_id = MATERIAL_MARBLE_LIGHT;
			}
			else if (material_hit(obj1))
			{
				float3 mp = geometry.pos * real(3.f);
				material_output.material_position.x = mp.x;
				material_output.material_position.y = mp.y;
				material_output.material_position.z = mp.z;
				material_output.material_id = MATERIAL_MARBLE_DARK;
			}
			else if (material_hit(wood))
			{
				float3 mp = float3(geometry.pos.x, geometry.pos.z, geometry.pos.y) * real(2.f);
				material_output.material_position.x = mp.x;
				material_output.material_position.y = mp.y;
				material_output.material_position.z = mp.z;
				material_output.material_id = MATERIAL_WOOD;
			}
			else if (material_hit(fire))
			{
				float3 mp = torch_pos * real(3.f) - float3(real(0.f), F.stime * real(3.f), real(0.f));
				material_output.material_position.x = mp.x;
				material_output.material_position.y = mp.y;
				material_output.material_position.z = mp.z;
				material_output.material_id = MATERIAL_FIRE;
			}
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_fractal.hlsl -------------------------------------------------
struct SceneFractal
{
	static const char *name() { return "fractal"; }
	// :6-15 -- x biggest, then y, then z
	static float3 sort_components(float3 vec)
	{
		real t;
		if (vec.z > vec.y) { t = vec.y; vec.y = vec.z; vec.z = t; }
		if (vec.y > vec.x) { t = vec.x; vec.x = vec.y; vec.y = t; }
		if (vec.z > vec.y) { t = vec.y; vec.y = vec.z; vec.z = t; }
		return vec;
	}
	// fractal_pos.yxz = sort_components(fractal_pos.yxz)  (:40,47; swizzled l-value, a-T.7)
	static void sort_yxz(float3 &p)
	{
		float3 s = sort_components(float3(p.y, p.x, p.z));
		p.y = s.x;
		p.x = s.y;
		p.z = s.z;
	}
	// :17-67
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);

		real size = 1.f;
		float3 fractal_pos = geometry.pos - float3(real(0.f), real(1.f), real(0.f));
		real g = 0.7f;

		real fractal = 1e30f;
		real scale = 1.f;
		real iters_needed = 0.f;

		for (int i = 0; i < 8; ++i)
		{
			// r_div_const: same quotient; the census build also records numerators outside the domain on
			// which the kernels' constant division (by 3^i) is proven exact
			real new_d = r_div_const(sdBox(fractal_pos, size * real(0.5f)), scale);
			if (new_d < real(0.0001f) && fractal > real(0.0001f))
			{
				iters_needed = real((float)i);
			}
			fractal = r_min(fractal, new_d);

			fractal_pos = v_abs(fractal_pos);
			sort_yxz(fractal_pos);

			fractal_pos.y -= size * real(2.f) / real(3.f);

			fractal_pos.y += size / real(3.f);
			sort_yxz(fractal_pos);
			fractal_pos.y -= size / real(3.f);

			fractal_pos = fractal_pos * real(3.f);
			scale *= real(3.f);
		}

		if (geometry_step)
		{
			object_add(output_scene_distance, fractal);
		}
		else if (material_hit(fractal))
		{
			material_output.diffuse_color.x = real(0.9f);
			material_output.diffuse_color.y = g;
			material_output.diffuse_color.z = iters_needed * real(0.125f);
			set_rgb(material_output.specular_color, real(0.5f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_lense.hlsl ---------------------------------------------------
// scene variables in declaration order: xpos, ypos, zpos (:29-31), mixing (:85)
struct SceneLense
{
	static const char *name() { return "lense"; }
	enum { VAR_XPOS = 0, VAR_YPOS = 1, VAR_ZPOS = 2, VAR_MIXING = 3 };
	// :5-95
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		// lower background (:8-12)
		float3 background1_pos = geometry.pos - float3(real(0.f), real(-5.f), real(0.f));
		float2 r1 = opRepInf(float2(background1_pos.x, background1_pos.z), float2(real(3.f)));
		background1_pos.x = r1.x;
		background1_pos.z = r1.y;
		real background1_sphere = sdSphere(background1_pos, real(1.f));
		real background1_box = sdBox(background1_pos, real(1.f));
		real background1 = r_lerp(background1_sphere, background1_box, real(0.65f)) - real(0.1f);

		// upper background (:15-19)
		float3 background2_pos = geometry.pos - float3(real(0.f), real(5.f), real(0.f));
		float2 r2 = opRepInf(float2(background2_pos.x, background2_pos.z), float2(real(10.f)));
		background2_pos.x = r2.x;
		background2_pos.z = r2.y;
		real background2_sphere = sdSphere(background2_pos, real(1.f));
		real background2_box = sdBox(background2_pos, real(1.f));
		real background2 = r_lerp(background2_sphere, background2_box, real(0.65f)) - real(0.1f);

		// lense (:22-26)
		float3 lense_pos = v_abs(geometry.pos);
		lense_pos.z -= real(5.1f);
		real lense1 = sdSphere(lense_pos, real(5.f));
		real lense2 = sdSphere(geometry.pos, real(2.f));
		real lense = r_max(-lense1, lense2);

		// sphere (:29-33)
		real x = F.scene_var[VAR_XPOS];
		real y = F.scene_var[VAR_YPOS];
		real z = F.scene_var[VAR_ZPOS];
		float3 sphere_pos = geometry.pos - float3(x, y, z);
		real sphere = sdSphere(sphere_pos, real(2.f));

		// mirror (:36-40)
		float3 mirror_position = geometry.pos - float3(real(0.f), real(0.f), real(-5.f));
		float2 mr = opRotate(float2(mirror_position.x, mirror_position.z), F.stime * real(0.3f));
		mirror_position.x = mr.x;
		mirror_position.z = mr.y;
		real mirror = sdBox(mirror_position, float3(real(1.f), real(2.f), real(0.1f)));
		real mirror_frame = sdBox(mirror_position, float3(real(1.1f), real(2.1f), real(0.08f)));

		if (geometry_step)
		{
			object_add(output_scene_distance, background1);
			object_add(output_scene_distance, background2);
			object_add(output_scene_distance, lense);
			object_add(output_scene_distance, sphere);
			object_add(output_scene_distance, mirror);
			object_add(output_scene_distance, mirror_frame);
		}
		else
		{
			float2 cell_index = (float2(geometry.pos.x, geometry.pos.z) - float2(background1_pos.x, background1_pos.z)) / real(3.f);

			if (material_hit(background1))
			{
				float2 sc = v_sin(cell_index * real(0.3f)) * real(0.5f) + real(0.5f);
				float3 cell_color1 = float3(sc, real(1.f));
				float3 cell_color2 = (cell_index.x < real(0.01f)) ? float3(real(0.f), real(1.f), real(0.f)) : float3(real(0.f), real(0.f), real(1.f));
				float3 cell_color = lerp(cell_color1, cell_color2, real(0.25f));
				material_output.diffuse_color = float4(cell_color, real(1.f));
				set_rgb(material_output.specular_color, real(1.f));
				material_output.reflection_color = float3(real(0.5f));
			}
			else if (material_hit(background2))
			{
				material_output.diffuse_color = float4(real(1.f), real(0.5f), real(0.f), real(1.f));
				set_rgb(material_output.specular_color, real(1.f));
				material_output.reflection_color = float3(real(0.5f));
			}
			else if (material_hit(lense))
			{
				material_output.diffuse_color = float4(real(0.3f), real(0.3f), real(0.3f), real(1.f));
				material_output.refraction_color = float3(real(0.9f), real(0.9f), real(0.9f));
			}
			else if (material_hit(sphere))
			{
				material_output.diffuse_color = float4(real(1.f), real(0.2f), real(0.2f), real(1.f));
				material_output.emissive_color = float3(real(8.f), real(0.f), real(0.f));
				set_rgb(material_output.specular_color, real(1.f));
				material_output.reflection_color = float3(real(0.25f));
			}
			else if (material_hit(mirror))
			{
				material_output.diffuse_color = float4(real(0.1f), real(0.1f), real(0.1f), real(1.f));
				real mix_ratio = F.scene_var[VAR_MIXING];
				material_output.refraction_color = float3(mix_ratio);
				material_output.reflection_color = float3(real(1.f) - mix_ratio);
			}
			else if (material_hit(mirror_frame))
			{
				material_output.material_position.x = mirror_position.x;
				material_output.material_position.y = mirror_position.y;
				material_output.material_position.z = mirror_position.z;
				material_output.material_id = MATERIAL_WOOD;
			}
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	// :108-117 -- own copy of the sky (identical arithmetic to sky_color)
	static float3 map_background(const Frame &F, float3 dir, uint)
	{
		float2 rot = opRotate(float2(dir.x, dir.z), -F.stime * real(0.025f));
		dir.x = rot.x;
		dir.z = rot.y;
		real noiseval = turbulence(dir * float3(real(1.f), real(6.f), real(1.f)) * real(2.5f));
		float3 color1 = float3(real(43.f), real(164.f), real(247.f)) / real(255.f);
		float3 color2 = float3(real(212.f), real(224.f), real(238.f)) / real(255.f);
		float3 sky = lerp(color1, color2, noiseval) * real(1.2f);
		float3 horizon_color = float3(real(0.25f));
		return lerp(horizon_color, sky, r_saturate(dir.y * real(8.f) + real(0.125f)));
	}
};

// ---- scenes/sdf_scene_gems.hlsl ----------------------------------------------------
struct SceneGems
{
	static const char *name() { return "gems"; }
	// :5-35
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);

		float3 obj_pos = geometry.pos;
		float2 xz = opRotate(float2(obj_pos.x, obj_pos.z), F.stime * real(0.5f));
		real index = opRepAngle(xz, real(8.f));
		obj_pos.x = xz.x;
		obj_pos.z = xz.y;
		obj_pos.x -= real(1.f);
		obj_pos.y -= real(1.f);
		xz = float2(obj_pos.x, obj_pos.z);
		opRepAngle(xz, real(8.f));
		obj_pos.x = xz.x;
		obj_pos.z = xz.y;
		real plane1 = sdPlane(obj_pos - float3(real(0.1f), real(0.1f), real(0.f)), float3(real(0.707f), real(0.707f), real(0.f)));
		real plane2 = sdPlane(obj_pos, float3(real(0.707f), real(-0.707f), real(0.f)));
		real plane3 = sdPlane(obj_pos - float3(real(0.f), real(0.13f), real(0.f)), float3(real(0.f), real(1.f), real(0.f)));
		real gems = smax2(smax2(plane1, plane2, real(0.001f)), plane3, real(0.001f));

		if (geometry_step)
		{
			object_add(output_scene_distance, gems);
		}
		else if (material_hit(gems))
		{
			float3 ruby_color = float3(real(0.8f), real(0.1f), real(0.3f));
			float3 saph_color = float3(real(0.8f), real(0.7f), real(0.1f));
			float3 c = (r_frac(index * real(0.5f) + real(0.25f)) > real(0.5f)) ? ruby_color : saph_color;
			material_output.diffuse_color.x = c.x;
			material_output.diffuse_color.y = c.y;
			material_output.diffuse_color.z = c.z;
			set_rgb(material_output.specular_color, real(1.f));
			material_output.refraction_color = float3(real(0.5f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_light_shadows.hlsl -------------------------------------------
// BACKWARDS = false is the scene file as it stands.  BACKWARDS = true steps the five lights' phases the other way (time -= 2 pi / 5):
// a guess at the older version of this file that the reference's own screenshot Images/multi-lights.png was taken with -- there the
// lights stand where today's file puts them and wear the colours in reverse order (tests/test_reference_images_cpu.py); it exists for
// that comparison only ("light_shadows_backwards", oracle_api.cpp: test scenes).
template <bool BACKWARDS>
struct SceneLightShadowsT
{
	static const char *name() { return BACKWARDS ? "light_shadows_backwards" : "light_shadows"; }
	// :5-11
	static float3 color_from_index(uint index)
	{
		real h = real((float)index) / real(5.f);
		float3 color = HSVtoRGB(float3(h, real(1.f), real(1.f)));
		real brightness = RGBtoBrightness(color);
		return color / brightness;
	}
	// :13-62
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);

		float3 cube_pos = geometry.pos;
		float2 rep = opRepLim(float2(cube_pos.x, cube_pos.z), float2(real(2.f), real(2.f)), float2(real(3.f), real(3.f)));
		cube_pos.x = rep.x;
		cube_pos.z = rep.y;
		real cubes = sdBox(cube_pos - float3(real(0.f), real(1.f), real(0.f)), real(0.4f)) - real(0.1f);

		real time = F.stime * real(0.25f);

		real spheres[5];
		for (uint i = 0; i < 5; ++i)
		{
			time = BACKWARDS ? time - real(pi) * real(2.f) / real(5.f) : time + real(pi) * real(2.f) / real(5.f);
			real sphere_x = r_cos(time * real(1.f)) * real(-5.f);
			real sphere_y = (r_cos(time * real(2.f)) * real(-0.5f) + real(0.5f)) * real(2.f) + real(1.f);
			real sphere_z = r_sin(time * real(2.f)) * real(2.f);
			spheres[i] = sdSphere(geometry.pos - float3(sphere_x, sphere_y, sphere_z), real(0.2f));
		}

		if (geometry_step)
		{
			if (!march.is_shadow_pass)
			{
				object_add(output_scene_distance, spheres[0]);
				object_add(output_scene_distance, spheres[1]);
				object_add(output_scene_distance, spheres[2]);
				object_add(output_scene_distance, spheres[3]);
				object_add(output_scene_distance, spheres[4]);
			}
			object_add(output_scene_distance, cubes);
		}
		else
		{
			for (uint i = 0; i < 5; ++i)
			{
				if (material_hit(spheres[i]))
				{
					material_output.emissive_color = color_from_index(i);
				}
			}
			if (material_hit(cubes))
			{
				set_rgb(material_output.diffuse_color, real(0.65f));
				set_rgb(material_output.specular_color, real(0.75f));
			}
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	// :68-86
	static void map_light(const Frame &F, const GeometryInput &, LightOutput *output, real &)
	{
		real time = F.stime * real(0.25f);
		for (uint i = 0; i < 5; ++i)
		{
			time = BACKWARDS ? time - real(pi) * real(2.f) / real(5.f) : time + real(pi) * real(2.f) / real(5.f);
			real sphere_x = r_cos(time * real(1.f)) * real(-5.f);
			real sphere_y = (r_cos(time * real(2.f)) * real(-0.5f) + real(0.5f)) * real(2.f) + real(0.5f);
			real sphere_z = r_sin(time * real(2.f)) * real(2.f);

			output[i + 1].used = true;
			output[i + 1].pos.x = sphere_x;
			output[i + 1].pos.y = sphere_y;
			output[i + 1].pos.z = sphere_z;
			output[i + 1].extend = real(0.25f);
			output[i + 1].falloff = real(0.25f);
			output[i + 1].color = color_from_index(i) * real(0.5f);
		}
	}
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};
typedef SceneLightShadowsT<false> SceneLightShadows;

} // namespace orc
