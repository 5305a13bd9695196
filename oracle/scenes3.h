// oracle/scenes3.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates five more scene plugins:
//   Engine/shader/scenes/sdf_scene_{fractal2,shell,spiral,terrain,tiling}.hlsl
// and the tiling helpers of Engine/shader/sdf_materials.hlsl:33-140 (voronoi, truchet_band,
// braid) they use.
#pragma once
#include "scenes2.h"

namespace orc {

// HLSL atan(x), defined here through the deterministic atan2 as atan2(x, 1)
inline real r_atan(real a) { return r_atan2(a, real(1.f)); }

// ---- sdf_materials.hlsl:33-38 ------------------------------------------------------
inline float2 voronoi_cell_offset(float2 cell_index)
{
	real x = hashf((uint32_t)r_ftoi(cell_index.x + cell_index.y * real(217.743f)));
	real y = hashf((uint32_t)r_ftoi(cell_index.x + cell_index.y * real(217.743f) + real(2475.235f)));
	return float2(x, y) * real(2.f) - real(1.f);
}

// sdf_materials.hlsl:45-92: (cell id xy, distance to the closest centre, distance to the closest edge)
inline float4 voronoi(float2 uv, real max_offset)
{
	float2 cell_index = v_floor(uv);
	float2 cell_pos = (uv - cell_index) - real(0.5f);

	real l_center_min = 10.f;
	real l_edge_min = 10.f;
	float2 closest_cell_id = float2(real(0.f));
	float2 closest_cell = float2(real(0.f));
	for (int x = -1; x < 2; ++x)
		for (int y = -1; y < 2; ++y)
		{
			float2 offset = float2(real((float)x), real((float)y));
			float2 cell_id = cell_index + offset;
			float2 point_pos = offset + voronoi_cell_offset(cell_id) * max_offset;
			float2 center_vec = point_pos - cell_pos;
			real l_center = length(center_vec);
			if (l_center < l_center_min)
			{
				l_center_min = l_center;
				closest_cell_id = cell_id;
				closest_cell = point_pos;
			}
		}
	for (int x = -1; x < 2; ++x)
		for (int y = -1; y < 2; ++y)
		{
			float2 offset = float2(real((float)x), real((float)y));
			float2 cell_id = cell_index + offset;
			float2 point_pos = offset + voronoi_cell_offset(cell_id) * max_offset;
			float2 border_vec = (point_pos + closest_cell) * real(0.5f);
			real edge_dist = r_abs(dot(normalize(closest_cell - border_vec), cell_pos - border_vec));
			l_edge_min = r_min(edge_dist, l_edge_min);
		}
	return float4(closest_cell_id.x, closest_cell_id.y, l_center_min, l_edge_min);
}

// sdf_materials.hlsl:97-119
inline float4 truchet_band(float2 uv, real chance, real width, float2 miss_uv)
{
	float2 cell_index = v_floor(uv);
	float2 cell_pos = (uv - cell_index) - real(0.5f);

	real flip3 = r_step(r_frac((cell_index.x + cell_index.y) * real(0.5f) + real(0.25f)), real(0.5f)) * real(2.f) - real(1.f);
	real flip2 = r_step(hashf((uint32_t)r_ftoi(cell_index.x + cell_index.y * real(217.743f))), chance) * real(2.f) - real(1.f);
	cell_pos.y *= flip2;
	real flip1 = r_step(cell_pos.y, cell_pos.x) * real(2.f) - real(1.f);
	cell_pos = cell_pos * flip1;
	cell_pos = cell_pos + float2(real(-0.5f), real(0.5f));
	real len = length(cell_pos);

	if (r_abs(len - real(0.5f)) < width)
	{
		real a = (len - real(0.5f) + width) / (real(2.f) * width);
		real b = r_atan2(cell_pos.y, -cell_pos.x) / (real(3.1415926f) * real(0.5f));
		a = r_lerp(real(1.f) - a, a, flip2 * flip3 * real(0.5f) + real(0.5f));
		b = r_lerp(real(1.f) - b, b, flip3 * real(0.5f) + real(0.5f));
		return float4(cell_index.x, cell_index.y, a, b);
	}
	return float4(cell_index.x, cell_index.y, miss_uv.x, miss_uv.y);
}

// sdf_materials.hlsl:125-140
inline float4 braid(float2 uv, real width, real run_length, real run_flip, float2 miss_uv)
{
	float2 cell_index = v_floor(uv);
	float2 cell_pos = (uv - cell_index) - real(0.5f);

	real t = r_frac((cell_index.x + cell_index.y) / run_length) * run_length + real(0.5f);
	real flip = r_step(t, run_flip);

	cell_pos = lerp(cell_pos, float2(cell_pos.y, cell_pos.x), flip);
	float2 rel_pos = v_abs(cell_pos) / width;
	float2 overflow = v_step(real(1.f), rel_pos);
	cell_pos = lerp(cell_pos, float2(cell_pos.y, cell_pos.x), overflow.x);
	cell_pos = lerp(cell_pos, miss_uv, overflow.x * overflow.y);
	return float4(cell_index.x, cell_index.y, cell_pos.x, cell_pos.y);
}

// the sky with an adjustable cloud mix (scenes that carry their own copy of sky_color)
inline float3 sky_color_mix(float3 dir, real phase, real mix_scale, real mix_bias)
{
	float2 rot = opRotate(float2(dir.x, dir.z), -phase * real(0.025f));
	dir.x = rot.x;
	dir.z = rot.y;
	real noiseval = turbulence(dir * float3(real(1.f), real(6.f), real(1.f)) * real(2.5f));
	float3 color1 = float3(real(43.f), real(164.f), real(247.f)) / real(255.f);
	float3 color2 = float3(real(212.f), real(224.f), real(238.f)) / real(255.f);
	float3 sky = lerp(color1, color2, noiseval * mix_scale + mix_bias) * real(1.2f);
	return lerp(float3(real(0.25f)), sky, r_saturate(dir.y * real(8.f) + real(0.125f)));
}

// ---- scenes/sdf_scene_fractal2.hlsl ------------------------------------------------
// variable: slider (:37, declared but unused by the scene)
struct SceneFractal2
{
	// :18-24
	static real mod(real input, real lower, real upper)
	{
		real range = upper - lower;
		real reduced = (input - lower) / range;
		real fract = reduced - r_floor(reduced);
		return fract * range + lower;
	}
	// :26-90
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);

		real size = 1.f;
		float3 fractal_pos = geometry.pos - float3(real(0.f), real(1.f), real(0.f));
		float3 fractal_base_pos = fractal_pos;
		real fractal_slice = dot(fractal_base_pos, float3(real(1.f)));

		real fractal = 1e30f;
		real scale = 1.f;
		for (uint i = 0; i < 6; ++i)
		{
			real new_d = r_div_const(sdBox(fractal_pos, size * real(0.5f)), scale); // census: numerator domain of the kernels' div_c
			fractal = r_min(fractal, new_d);

			fractal_pos = v_abs(fractal_pos);
			SceneFractal::sort_yxz(fractal_pos);
			fractal_pos.y -= size * real(2.f) / real(3.f);
			// offset blocks (:60)
			fractal_pos.z -= r_step(size * real(0.5f) / real(3.f), fractal_pos.z) * size / real(3.f) * real(1.001f);
			fractal_pos.y += size / real(3.f);
			SceneFractal::sort_yxz(fractal_pos);
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
			real slice_size = 0.01f;
			real diff = fractal_slice - F.stime * real(0.5f) - snoise(fractal_base_pos) * real(0.5f);
			diff = mod(diff, real(-0.5f), real(0.5f));
			real colorize = r_saturate(slice_size - r_abs(diff)) / slice_size;

			real len = length(fractal_base_pos);
			float3 color1 = float3(real(1.f), real(0.8f), real(0.1f));
			float3 color2 = float3(real(0.8f), real(0.3f), real(0.1f)) * colorize * real(1.5f);
			float3 glow_color = float3(real(0.1f), real(0.5f), real(0.1f)) * r_saturate((real(0.6f) - len) * real(10.f));
			material_output.diffuse_color.x = color1.x;
			material_output.diffuse_color.y = color1.y;
			material_output.diffuse_color.z = color1.z;
			material_output.emissive_color = color2 + glow_color;
			set_rgb(material_output.specular_color, real(0.5f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	// :96-103
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &ambient_lighting_factor)
	{
		output[0].used = true;
		output[0].pos = float4(real(-1.f), real(-4.f), real(2.f), real(1.f));
		output[0].color = float3(real(1.f), real(1.f), real(1.f));
		ambient_lighting_factor = 0.1f;
	}
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_shell.hlsl ---------------------------------------------------
struct SceneShell
{
	// :45-83 (the scene carries its own copies of the checker helpers :5-43, same arithmetic
	// as sdf_common.hlsl:24-60)
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		real cube1 = sdBox(geometry.pos - float3(real(0.f), real(1.f), real(0.f)), real(0.5f));
		cube1 = opShell(cube1, real(0.f), real(0.3f));
		cube1 = opShell(cube1, real(-0.05f), real(0.05f));
		real cut1 = sdPlane(geometry.pos, float3(real(-1.f), real(0.f), real(0.f)));
		cube1 = r_max(cube1, -cut1);
		real floor1 = sdPlaneFast(geometry.pos, geometry.dir, float3(real(0.f), real(1.f), real(0.f)));

		if (geometry_step)
		{
			object_add(output_scene_distance, cube1);
			object_add(output_scene_distance, floor1);
		}
		else
		{
			if (material_hit(cube1))
			{
				material_output.diffuse_color = float4(real(0.6f), real(0.5f), real(0.2f), real(1.f));
				set_rgb(material_output.specular_color, real(0.5f));
				material_output.reflection_color = float3(real(0.15f));
			}
			else if (material_hit(floor1))
			{
				float3 offset_right = geometry.right_ray_offset * geometry.camera_distance;
				float3 offset_bottom = geometry.bottom_ray_offset * geometry.camera_distance;
				float3 color = total_tile_color(geometry.pos, geometry.dir.xyz(), offset_right, offset_bottom);
				material_output.diffuse_color = float4(color, real(1.f));
				set_rgb(material_output.specular_color, real(1.f));
			}
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	// :89-94
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &)
	{
		output[0].used = true;
		output[0].pos = float4(real(-1.f), real(-1.f), real(2.f), real(1.f));
		output[0].color = float3(real(1.f), real(1.2f), real(1.f));
	}
	// :96-105
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color_mix(dir, F.stime, real(1.f), real(0.f)); }
};

// ---- scenes/sdf_scene_spiral.hlsl --------------------------------------------------
struct SceneSpiral
{
	// :11-40
	static real sdSpiral(float3 pos, real r1, real h, real r2, real angle_start, real angle_end)
	{
		real height_per_rotation = real(tau) * h / (angle_end - angle_start);
		real rel_height = height_per_rotation * r_atan2(pos.z, pos.x) / real(tau);
		real start_height = height_per_rotation * angle_start / real(tau);

		real offset_pos_y = pos.y;
		real height_diff = r_clamp(offset_pos_y, height_per_rotation * real(0.5f), h - height_per_rotation * real(0.5f));
		height_diff -= rel_height - start_height;
		offset_pos_y -= rel_height - start_height;
		real closest_height = r_round(height_diff / height_per_rotation) * height_per_rotation;

		real axial_dist = closest_height - offset_pos_y;
		real radial_length = length(float2(pos.x, pos.z));
		real radial_dist = radial_length - r1;
		float2 ab = float2(radial_dist, axial_dist);
		real body_length = length(ab) - r2;

		float3 cap1 = float3(r1 * r_cos(angle_start), real(0.f), r1 * r_sin(angle_start));
		float3 cap2 = float3(r1 * r_cos(angle_end), h, r1 * r_sin(angle_end));
		return r_min(body_length, r_min(length(pos - cap1) - r2, length(pos - cap2) - r2));
	}
	// :42-88
	static void map(const Frame &F, const GeometryInput &geometry_in, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		GeometryInput geometry = geometry_in;
		real speed = 1.5f;
		real total_x = F.stime * speed;
		real width = 4.f, height = 6.f, spring_length = 3.f, pen = 2.f;

		real arc_pos = r_frac(total_x / width);
		real x = arc_pos * width;
		real y = arc_pos * (real(1.f) - arc_pos) * real(4.f) * height;
		real y_top = y - pen + spring_length;
		real y_bottom = r_max(y - pen, real(0.f));
		real dydx = (real(1.f) - real(2.f) * arc_pos) * real(4.f) * height / width;
		real spring_angle = -r_atan(dydx) - real(pi) * real(0.5f);
		real spring_s = r_sin(spring_angle), spring_c = r_cos(spring_angle);

		float3 spring_pos = geometry.pos;
		spring_pos.y -= (y_top + y_bottom) * real(0.5f) + real(0.1f);
		spring_length = y_top - y_bottom;
		{
			real sx = spring_pos.x * spring_c - spring_pos.y * spring_s;
			real sy = spring_pos.x * spring_s + spring_pos.y * spring_c;
			spring_pos.x = sx;
			spring_pos.y = sy;
		}
		real obj = sdSpiral(spring_pos + float3(real(0.f), spring_length * real(0.5f), real(0.f)), real(1.f), spring_length, real(0.1f), real(0.f),
			real(4.5f) * real(tau)) * real(0.98f);

		geometry.pos.x += x;
		map_groundplane(geometry, material_output, geometry_step, output_scene_distance);

		if (geometry_step)
		{
			object_add(output_scene_distance, obj);
		}
		else if (material_hit(obj))
		{
			set_rgb(material_output.diffuse_color, real(0.5f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_terrain.hlsl -------------------------------------------------
// variable: levels (:40)
struct SceneTerrain
{
	// :5-8
	static real fast_noise(float3 p)
	{
		return r_frac(r_sin(dot(p, float3(real(12.9898f), real(78.233f), real(34.531247f)))) * real(43758.5453f));
	}
	// :10-16
	static real sdSphereCorner(float3 center, float3 pos, float3 offset)
	{
		float3 sphere_pos = center + offset;
		real rad = r_lerp(real(0.f), real(0.3f), fast_noise(sphere_pos));
		return sdSphere(pos - sphere_pos, rad);
	}
	// :18-32
	static real sdBase(float3 pos)
	{
		float3 ipos = v_floor(pos);
		real a = sdSphereCorner(ipos, pos, float3(real(0.f), real(0.f), real(0.f)));
		real b = sdSphereCorner(ipos, pos, float3(real(0.f), real(0.f), real(1.f)));
		real c = sdSphereCorner(ipos, pos, float3(real(0.f), real(1.f), real(0.f)));
		real d = sdSphereCorner(ipos, pos, float3(real(0.f), real(1.f), real(1.f)));
		real e = sdSphereCorner(ipos, pos, float3(real(1.f), real(0.f), real(0.f)));
		real f = sdSphereCorner(ipos, pos, float3(real(1.f), real(0.f), real(1.f)));
		real g = sdSphereCorner(ipos, pos, float3(real(1.f), real(1.f), real(0.f)));
		real h = sdSphereCorner(ipos, pos, float3(real(1.f), real(1.f), real(1.f)));
		return r_min(r_min(r_min(a, b), r_min(c, d)), r_min(r_min(e, f), r_min(g, h)));
	}
	// :34-56; mul(mat, p) = (dot(row0, p), dot(row1, p), dot(row2, p))
	static real sdFbm(const Frame &F, float3 p, real d)
	{
		const float3 r0 = float3(real(0.00f), real(1.60f), real(1.20f));
		const float3 r1 = float3(real(-1.60f), real(0.72f), real(-0.96f));
		const float3 r2 = float3(real(-1.20f), real(-0.96f), real(1.28f));
		real s = 1.0f;
		const int levels = r_ftoi(F.scene_var[0]);
		for (int i = 0; i < levels; i++)
		{
			real n = s * sdBase(p);
			n = smax2(n, d - real(0.1f) * s, real(0.3f) * s);
			d = smin(n, d, real(0.3f) * s);
			p = float3(dot(r0, p), dot(r1, p), dot(r2, p));
			float2 rot = opRotate(float2(p.x, p.z), real(1.f));
			p.x = rot.x;
			p.z = rot.y;
			s = real(0.5f) * s;
		}
		return d;
	}
	// :58-78 (no ground plane)
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		real box = sdBox(geometry.pos, float3(real(5.f), real(5.f), real(5.f)));
		real plane = sdPlane(geometry.pos, float3(real(0.f), real(1.f), real(0.f)));
		real obj = sdFbm(F, geometry.pos, plane);
		obj = r_max(obj, box);
		if (geometry_step)
		{
			object_add(output_scene_distance, obj);
		}
		else if (material_hit(obj))
		{
			material_output.diffuse_color.x = real(0.8f);
			material_output.diffuse_color.y = real(0.8f);
			material_output.diffuse_color.z = real(0.8f);
			set_rgb(material_output.specular_color, real(0.5f));
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &) { default_directional_light(output); }
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

// ---- scenes/sdf_scene_tiling.hlsl --------------------------------------------------
// variables in order of appearance: m1, m2, width, run_length, run_flip (:60-64),
// flip_chance, truchet_width (:76-77)
struct SceneTiling
{
	enum { V_M1 = 0, V_M2, V_WIDTH, V_RUN_LENGTH, V_RUN_FLIP, V_FLIP_CHANCE, V_TRUCHET_WIDTH };
	// :26-33
	static float3 random_color(float2 cell_index)
	{
		real r = hashf((uint32_t)r_ftoi(cell_index.x + cell_index.y * real(217.743f)));
		real g = hashf((uint32_t)r_ftoi(cell_index.x + cell_index.y * real(217.743f) + real(2475.235f)));
		real b = hashf((uint32_t)r_ftoi(cell_index.x + cell_index.y * real(217.743f) + real(824.213f)));
		real maxval = r_max(r_max(r, g), b);
		return float3(r, g, b) / maxval;
	}
	// :35-140
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		float3 cable_pos = geometry.pos - float3(real(4.f), real(4.f), real(4.f));
		real cable_radius = 0.1f;
		real pane1 = sdBox(geometry.pos - float3(real(-4.f), real(4.f), real(0.f)), float3(real(1.f), real(2.f), real(0.05f)));
		real pane2 = sdBox(geometry.pos - float3(real(0.f), real(4.f), real(0.f)), float3(real(1.f), real(2.f), real(0.05f)));
		real pane3 = sdBox(geometry.pos - float3(real(4.f), real(4.f), real(0.f)), float3(real(1.f), real(2.f), real(0.05f)));
		real cable = sdCappedCylinder(cable_pos, real(2.f), cable_radius);
		real floor1 = sdPlaneFast(geometry.pos, geometry.dir, float3(real(0.f), real(1.f), real(0.f)));

		if (geometry_step)
		{
			object_add(output_scene_distance, pane1);
			object_add(output_scene_distance, pane2);
			object_add(output_scene_distance, pane3);
			object_add(output_scene_distance, cable);
			object_add(output_scene_distance, floor1);
		}
		else
		{
			float2 uv = float2(geometry.pos.x, geometry.pos.y);
			real m1 = F.scene_var[V_M1], m2 = F.scene_var[V_M2], width = F.scene_var[V_WIDTH];
			real run_length = F.scene_var[V_RUN_LENGTH], run_flip = F.scene_var[V_RUN_FLIP];

			if (material_hit(pane1))
			{
				float4 v = voronoi(uv * real(5.f), real(0.45f));
				float3 color = (v.w > real(0.05f)) ? random_color(float2(v.x, v.y)) * real(1.1f) : float3(real(0.25f), real(0.25f), real(0.25f));
				material_output.diffuse_color = float4(color, real(1.f));
				set_rgb(material_output.specular_color, real(0.4f));
				material_output.specular_color.w = real(20.f);
			}
			else if (material_hit(pane2))
			{
				real flip_chance = F.scene_var[V_FLIP_CHANCE];
				real twidth = F.scene_var[V_TRUCHET_WIDTH];
				float2 uv_prime = opAB2UV(uv);
				float4 truchet = truchet_band(uv_prime * real(3.f), flip_chance, twidth, float2(real(0.f), real(-1.f)));
				real green = (truchet.w < real(0.f)) ? real(0.f) : r_sin(truchet.w * real(2.f) * real(pi) * real(5.f) + F.stime * real(2.f)) * real(0.5f) + real(0.5f);
				material_output.diffuse_color = float4(real(0.f), green * green, real(0.f), real(1.f));
			}
			else if (material_hit(pane3))
			{
				uv = opAB2UV(uv * real(5.f));
				uv = float2(uv.x * m1 + uv.y * m2, uv.x * m2 + uv.y * m1);
				float4 pattern = braid(uv, width, run_length, run_flip, float2(real(-2.f), real(0.f)));
				real grey = r_step(real(-1.f), pattern.z) * (r_cos(pattern.z * real(50.f)) * real(0.5f) + real(0.5f));
				float3 color = float3(grey * grey);
				material_output.diffuse_color = float4(color * real(0.8f), real(1.f));
			}
			else if (material_hit(cable))
			{
				real angle = r_atan2(cable_pos.z, cable_pos.x);
				float2 uv_round = float2(angle * cable_radius, cable_pos.y);
				uv_round = opAB2UV(uv_round * real(8.f));
				uv_round = float2(uv_round.x * m1 + uv_round.y * m2, uv_round.x * m2 + uv_round.y * m1);
				float4 pattern = braid(uv_round, width, run_length, run_flip, float2(real(-2.f), real(0.f)));
				real grey = r_step(real(-1.f), pattern.z) * (r_cos(pattern.z * real(50.f)) * real(0.5f) + real(0.5f));
				float3 color = float3(grey * grey);
				material_output.diffuse_color = float4(color * real(0.9f), real(1.f));
			}
			else if (material_hit(floor1))
			{
				float3 offset_right = geometry.right_ray_offset * geometry.camera_distance;
				float3 offset_bottom = geometry.bottom_ray_offset * geometry.camera_distance;
				float3 color = total_tile_color(geometry.pos, geometry.dir.xyz(), offset_right, offset_bottom);
				material_output.diffuse_color = float4(color, real(1.f));
				set_rgb(material_output.specular_color, real(0.5f));
			}
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	// :146-151
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &)
	{
		output[0].used = true;
		output[0].pos = float4(real(-1.f), real(-1.f), real(2.f), real(1.f));
		output[0].color = float3(real(1.f), real(1.2f), real(1.f));
	}
	// :153-162 (cloud mix noiseval * 0.8 + 0.2)
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color_mix(dir, F.stime, real(0.8f), real(0.2f)); }
};

} // namespace orc
