// oracle/scenes2.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates nine more of the reference's scene plugins:
//   Engine/shader/scenes/sdf_scene_{cube,gyroid,basic_transparency,basic_clouds,
//   coordinate_material,distortion,table,sierpinski,neon}.hlsl
#pragma once
#include "scenes.h"

namespace orc {

// HLSL tan(x): the compiler expands it to sin(x) / cos(x)
inline real r_tan(real a) { return r_sin(a) / r_cos(a); }

// ---- scenes/sdf_scene_cube.hlsl ----------------------------------------------------
// variables in order of appearance: size, xpos, ypos, zpos (:9-12), red, green, blue (:25-27)
struct SceneCube
{
	enum { V_SIZE = 0, V_X, V_Y, V_Z, V_RED, V_GREEN, V_BLUE };
	// :5-33
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		real size = F.scene_var[V_SIZE];
		real x = F.scene_var[V_X], y = F.scene_var[V_Y], z = F.scene_var[V_Z];
		real cube = sdBox(geometry.pos - float3(real(0.f), real(1.f), real(0.f)) - float3(x, y, z), size);
		if (geometry_step)
		{
			object_add(output_scene_distance, cube);
		}
		else if (material_hit(cube))
		{
			material_output.diffuse_color = float4(F.scene_var[V_RED], F.scene_var[V_GREEN], F.scene_var[V_BLUE], real(1.f));
			set_rgb(material_output.specular_color, real(0.5f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_gyroid.hlsl --------------------------------------------------
struct SceneGyroid
{
	// :5-8
	static real sdGyroid(float3 p)
	{
		return dot(float3(r_sin(p.x), r_sin(p.y), r_sin(p.z)), float3(r_cos(p.z), r_cos(p.x), r_cos(p.y)));
	}
	// :10-31 (no ground plane in this scene)
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		real gyroid = sdGyroid(geometry.pos * real(7.f)) / real(14.f);
		gyroid = r_abs(gyroid) - real(0.01f);
		real box = sdBox(geometry.pos, float3(real(1.f), real(1.f), real(1.f)));
		real obj = r_max(gyroid, box);
		if (geometry_step)
		{
			object_add(output_scene_distance, obj);
		}
		else if (material_hit(obj))
		{
			material_output.diffuse_color.x = real(0.9f);
			material_output.diffuse_color.y = real(0.7f);
			material_output.diffuse_color.z = real(0.2f);
			set_rgb(material_output.specular_color, real(0.5f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_basic_transparency.hlsl --------------------------------------
struct SceneBasicTransparency
{
	// :6-39
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		const float3 size = float3(real(1.f), real(1.f), real(0.1f));
		real box1 = sdBox(geometry.pos - float3(real(0.f), real(2.f), real(-1.f)), size);
		real box2 = sdBox(geometry.pos - float3(real(0.f), real(2.f), real(0.f)), size);
		real box3 = sdBox(geometry.pos - float3(real(0.f), real(2.f), real(1.f)), size);
		real transparent_box1 = sdBox(march.last_transparent_pos - float3(real(0.f), real(2.f), real(-1.f)), size);
		real transparent_box2 = sdBox(march.last_transparent_pos - float3(real(0.f), real(2.f), real(0.f)), size);
		real transparent_box3 = sdBox(march.last_transparent_pos - float3(real(0.f), real(2.f), real(1.f)), size);
		if (geometry_step)
		{
			object_add_transparent(output_scene_distance, march, box1, transparent_box1);
			object_add_transparent(output_scene_distance, march, box2, transparent_box2);
			object_add_transparent(output_scene_distance, march, box3, transparent_box3);
		}
		else
		{
			if (material_hit(box1))
				material_output.diffuse_color = float4(real(0.9f), real(0.9f), real(0.f), real(0.3f));
			else if (material_hit(box2))
				material_output.diffuse_color = float4(real(0.f), real(0.9f), real(0.9f), real(0.3f));
			else if (material_hit(box3))
				material_output.diffuse_color = float4(real(0.9f), real(0.f), real(0.9f), real(0.3f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_basic_clouds.hlsl --------------------------------------------
// variable: offset (:23)
struct SceneBasicClouds
{
	// :6-34
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		float3 cloud_pos = geometry.pos - float3(real(0.f), real(5.f), real(0.f));
		real cloud = sdBox(cloud_pos, float3(real(2.f), real(0.5f), real(2.f)));
		real transparent_cloud = sdBox(march.last_transparent_pos - float3(real(0.f), real(5.f), real(0.f)), float3(real(2.f), real(0.5f), real(2.f)));
		if (geometry_step)
		{
			object_add_transparent(output_scene_distance, march, cloud, transparent_cloud);
		}
		else if (material_hit(cloud))
		{
			real thickness = real(0.f) + F.scene_var[0];
			for (uint i = 0; i < 5; ++i)
			{
				float3 sample_pos = cloud_pos + geometry.dir.xyz() * real(0.5f) * real((float)i);
				thickness += turbulence(sample_pos);
			}
			thickness = r_saturate(thickness);
			float3 color = real(1.f) - float3(thickness * real(0.2f));
			material_output.diffuse_color = float4(color, thickness);
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	// :40-45
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &)
	{
		output[0].used = true;
		output[0].pos = float4(real(-1.f), real(-4.f), real(1.5f), real(1.f));
		output[0].color = float3(real(1.f), real(1.f), real(1.f));
	}
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_coordinate_material.hlsl -------------------------------------
// variables: boxoffset (:25), spherical (:41), thres (:48)
struct SceneCoordinateMaterial
{
	// :6-12
	static float3 cartesian2spherical(float3 pos)
	{
		real r = length(pos);
		real theta = r_atan2(pos.y, length(float2(pos.x, pos.z)));
		real phi = r_atan2(pos.z, pos.x);
		return float3(r, theta, phi);
	}
	// :14-19
	static void cartesian2spherical(float3 pos, float3 norm, float3 &out_pos, float3 &out_norm)
	{
		float3 p = cartesian2spherical(pos);
		float3 offset = cartesian2spherical(pos + norm * real(0.01f));
		out_pos = p;
		out_norm = normalize(offset - p);
	}
	// :21-55
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		real box_offset = F.scene_var[0];
		real sphere = sdSphere(geometry.pos - float3(real(0.f), real(2.f), real(0.f)), real(2.f));
		real box = sdBox(geometry.pos - float3(real(-1.f), real(3.f) + box_offset, real(-1.f)), real(1.f));
		sphere = r_max(sphere, -box);
		if (geometry_step)
		{
			object_add(output_scene_distance, sphere);
		}
		else if (material_hit(sphere))
		{
			float3 pos = (geometry.pos - float3(real(0.f), real(2.f), real(0.f))) * real(2.f);
			float3 norm = material_input.obj_normal;
			real use_spherical = F.scene_var[1];
			if (use_spherical > real(0.5f))
			{
				cartesian2spherical(pos, norm, pos, norm);
				pos = pos * float3(real(1.f), real(8.f) / real(pi), real(8.f) / real(pi));
			}
			real sel = coordinate_material(pos, norm, real(0.02f));
			float3 color = (sel > F.scene_var[2]) ? float3(real(1.f), real(0.f), real(0.f)) : float3(real(0.8f), real(0.8f), real(0.8f));
			material_output.diffuse_color.x = color.x;
			material_output.diffuse_color.y = color.y;
			material_output.diffuse_color.z = color.z;
			set_rgb(material_output.specular_color, real(0.25f));
			material_output.specular_color.w = real(100.f);
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_distortion.hlsl ----------------------------------------------
struct SceneDistortion
{
	// :10-14
	static real distort(real obj, real val, real lip, real h)
	{
		real actual_distance = (obj - val) / r_sqrt(real(1.f) + lip * lip);
		return r_lerp(actual_distance, obj - h, r_saturate(obj / h - real(1.f)));
	}
	// :16-51
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		float3 box_pos = geometry.pos - float3(real(0.f), real(1.5f), real(0.f));
		real scaled_pos = box_pos.y * real(2.5f) - real(1.25f);
		real index = scaled_pos - r_floor(scaled_pos);
		real offset = r_step(real(0.5f), index);
		real val_x = real(1.f) - r_pow(r_saturate(r_sin((box_pos.x + offset * real(0.2f)) * real(pi) * real(5.f))), real(10.f));
		real val_y = real(1.f) - r_pow(r_abs(r_sin(box_pos.y * real(pi) * real(5.f))), real(10.f));
		real n = turbulence(box_pos * real(7.5f));
		real v = r_min(val_x, val_y);
		v = r_lerp(v * real(0.8f), v, n);
		real height = 0.025f;
		real lip = 2.f;
		real box = sdBox(box_pos, float3(real(1.f), real(1.f), real(0.1f)));
		box = distort(box, v * height, lip * height, height);
		if (geometry_step)
		{
			object_add(output_scene_distance, box);
		}
		else if ((box - real(0.1f)) < real(dist_eps))
		{
			float3 color_wall1 = float3(real(0.8f), real(0.2f), real(0.2f));
			float3 color_wall2 = float3(real(0.5f), real(0.1f), real(0.1f));
			float3 color_gap = float3(real(0.5f), real(0.5f), real(0.5f));
			float3 color_wall = lerp(color_wall2, color_wall1, n);
			float3 color = (v < real(0.15f)) ? color_gap : color_wall;
			material_output.diffuse_color.x = color.x;
			material_output.diffuse_color.y = color.y;
			material_output.diffuse_color.z = color.z;
			set_rgb(material_output.specular_color, real(0.125f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_table.hlsl ---------------------------------------------------
struct SceneTable
{
	// :5-54 (constants :5-9)
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		const real leg_width = 0.05f, leg_height = 0.7f, leg_distance = 1.f, plate_size = 1.2f, plate_height = 0.0175f;
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		float3 p = geometry.pos;
		p.y -= real(0.4f);
		float3 abspos = v_abs(p);
		real legs = sdBox(abspos - float3(leg_distance, leg_height * real(0.5f), leg_distance), float3(leg_width, leg_height * real(0.5f), leg_width));
		real plate = sdBox(p - float3(real(0.f), leg_height + real(0.025f), real(0.f)), float3(plate_size, plate_height, plate_size)) - real(0.025f);

		real vase1 = sdSphere(p - float3(real(0.f), leg_height + real(0.15f), real(0.f)), real(0.2f));
		real vase2 = sdSphere(p - float3(real(0.f), leg_height + real(0.45f), real(0.f)), real(0.17f));
		real vase3 = sdSphere(p - float3(real(0.f), leg_height + real(0.72f), real(0.f)), real(0.15f));
		real vase_cut1 = sdPlane(p - float3(real(0.f), leg_height + real(0.615f), real(0.f)), float3(real(0.f), real(1.f), real(0.f)));
		real vase_cut2 = sdPlane(p - float3(real(0.f), leg_height + real(0.1f), real(0.f)), float3(real(0.f), real(-1.f), real(0.f)));
		real vase_body = r_max(smin(smin(vase1, vase2, real(0.05f)), vase3, real(0.025f)), vase_cut2);
		real vase = r_max(r_max(vase_body, vase_cut1), -vase_body - real(0.01f));

		if (geometry_step)
		{
			object_add(output_scene_distance, plate);
			object_add(output_scene_distance, legs);
			object_add(output_scene_distance, vase);
		}
		else
		{
			if (material_hit(plate))
			{
				real step = r_floor((p.x + real(1.25f)) * real(4.f)) / real(8.f);
				float3 mp = p + float3(p.z * real(0.2f), step, real(0.f));
				material_output.material_position.x = mp.x;
				material_output.material_position.y = mp.y;
				material_output.material_position.z = mp.z;
				material_output.material_id = MATERIAL_WOOD;
			}
			else if (material_hit(legs))
			{
				material_output.material_position.x = p.x;
				material_output.material_position.y = p.z;
				material_output.material_position.z = p.y;
				material_output.material_id = MATERIAL_WOOD;
			}
			else if (material_hit(vase))
			{
				float3 mp = p * real(4.f);
				material_output.material_position.x = mp.x;
				material_output.material_position.y = mp.y;
				material_output.material_position.z = mp.z;
				material_output.material_id = MATERIAL_MARBLE_DARK;
			}
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_sierpinski.hlsl ----------------------------------------------
struct SceneSierpinski
{
	// :5-37
	static real sdSierpinski(float3 p, uint depth)
	{
		float3 a1 = float3(real(0.f), real(1.f), real(0.f));
		float3 a2 = float3(real(-0.7f), real(0.f), real(-0.5f));
		float3 a3 = float3(real(0.7f), real(0.f), real(-0.5f));
		float3 a4 = float3(real(0.f), real(0.f), real(0.7f));
		real scale = 2.f;
		for (uint iter = 0; iter < depth; ++iter)
		{
			float3 c = a1;
			real dist = length(p - a1);
			real d = length(p - a2);
			if (d < dist) { c = a2; dist = d; }
			d = length(p - a3);
			if (d < dist) { c = a3; dist = d; }
			d = length(p - a4);
			if (d < dist) { c = a4; dist = d; }
			p = scale * p - c * (scale - real(1.f));
		}
		return length(p) / r_pow(scale, real((float)depth)) - real(0.002f);
	}
	// :39-57
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		real obj = sdSierpinski(geometry.pos - float3(real(0.f), real(1.f), real(0.f)), 10);
		if (geometry_step)
		{
			object_add(output_scene_distance, obj);
		}
		else if (material_hit(obj))
		{
			material_output.diffuse_color.x = real(0.9f);
			material_output.diffuse_color.y = real(0.7f);
			material_output.diffuse_color.z = real(0.2f);
			set_rgb(material_output.specular_color, real(0.5f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_neon.hlsl ----------------------------------------------------
// variables: r1, r2, spacing (:22-24), red, green, blue (:43-45)
struct SceneNeon
{
	// :5-16
	static real sdRingsphere(float3 pos, real spacing, real r1, real r2)
	{
		float3 hit_point = normalize(pos) * r1;
		real y = hit_point.y;
		real x = length(float2(hit_point.x, hit_point.z));
		real angle = r_atan2(y, x);
		angle = r_round(angle / spacing) * spacing;
		hit_point.y = r_tan(angle) * x;
		hit_point = normalize(hit_point) * r1;
		return length(pos - hit_point) - r2;
	}
	// :18-62
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
		real r1 = F.scene_var[0], r2 = F.scene_var[1], spacing = F.scene_var[2];
		real ringsphere = sdRingsphere(geometry.pos - float3(real(0.f), real(2.f), real(0.f)), spacing, r1, r2);

		float3 mirror_pos = geometry.pos - float3(real(0.f), real(2.f), real(2.75f));
		float2 rot = opRotate(float2(mirror_pos.x, mirror_pos.z), real(0.3f));
		mirror_pos.x = rot.x;
		mirror_pos.z = rot.y;
		real mirror = sdBox(mirror_pos, float3(real(1.f), real(1.7f), real(0.05f)));
		real mirror_border = sdBox(mirror_pos, float3(real(1.05f), real(1.75f), real(0.04f)));

		if (geometry_step)
		{
			object_add(output_scene_distance, ringsphere);
			object_add(output_scene_distance, mirror);
			object_add(output_scene_distance, mirror_border);
		}
		else
		{
			if (material_hit(ringsphere))
			{
				real r = F.scene_var[3], g = F.scene_var[4], b = F.scene_var[5];
				material_output.emissive_color = float3(r, g, b);
				float3 half_c = float3(r, g, b) / real(2.f);
				material_output.diffuse_color.x = half_c.x;
				material_output.diffuse_color.y = half_c.y;
				material_output.diffuse_color.z = half_c.z;
				set_rgb(material_output.specular_color, real(0.5f));
			}
			else if (material_hit(mirror))
			{
				material_output.reflection_color = float3(real(0.8f));
				set_rgb(material_output.specular_color, real(0.1f));
			}
			else if (material_hit(mirror_border))
			{
				material_output.diffuse_color.x = real(0.5f);
				material_output.diffuse_color.y = real(0.5f);
				material_output.diffuse_color.z = real(0.5f);
				set_rgb(material_output.specular_color, real(0.5f));
			}
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

} // namespace orc
