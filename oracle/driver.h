// oracle/driver.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates the raymarch driver Engine/shader/pshader_sdf.hlsl: debug-plane helpers
// (:86-109), map_geometry (:111-135), map_material (:137-162), grad (:164-177),
// march_ray (:179-220), find_next_ray/find_free_ray (:222-246) and ps_main (:260-639),
// plus the pixel -> NDC mapping of the full-screen quad (vshader.hlsl:12-16,
// FullscreenQuad.cpp:52-58).
//
// The compile-time limits ITER_COUNT/BOUNCE_COUNT/RAY_COUNT/LIGHT_COUNT/RANGE and the
// default max_cost are read from Frame (defaults = reference values).
#pragma once
#include "sdf_lib.h"

namespace orc {

enum { INVALID_DEPTH = 1000000 }; // pshader_sdf.hlsl:58

// pshader_sdf.hlsl:40-51
struct Ray
{
	float3 pos;
	float3 dir;
	float3 contribution;
	real inside_sign;
	float3 last_transparent_pos;
	bool has_transparent;
	real shadow_range;
	bool is_shadow_ray;
	uint depth;
};

// per-pixel statistics the tests and the bench use (ray := one
// iteration of the bounce loop)
struct PixelStats
{
	uint rays;       // bounce-loop iterations
	uint march_evals; // map_geometry calls made by march_ray
	uint hits;       // march_ray returned true
};

// :86-93
inline float3 get_debug_plane_point(const Frame &F) { return float3(F.debug_x, F.debug_y, F.debug_z); }
// :95-104
inline float3 get_debug_plane_normal(const Frame &F)
{
	float3 n = float3(F.debug_nx, F.debug_ny, F.debug_nz);
	return any(n) ? normalize(n) : float3(real(0.f));
}
// :106-109
inline bool debug_show_objects(const Frame &F) { return any(F.show_objects); }

// :111-135
// EXTENSION, not in the reference (SURVEY.md 8d cfg 5: "8 lights"): slots 1..n of the light table
// are overwritten with point lights orbiting at height 3, radius 5:
//   phi_i = stime * 0.25 + i * (2 pi / 7), pos = (5 cos phi_i, 3, 5 sin phi_i), extend .25,
//   falloff .25, colour = the hue wheel of scenes/sdf_scene_light_shadows.hlsl:5-11 at h = i / 7,
//   normalised to unit brightness, times 0.5.
inline void extension_lights(const Frame &F, LightOutput *output)
{
	for (int i = 1; i <= F.extension_lights && i < MAX_LIGHT_COUNT; ++i)
	{
		const real phi = F.stime * real(0.25f) + real((float)i) * real(6.28318530718f / 7.f);
		const real h = real((float)i) / real(7.f);
		float3 color = HSVtoRGB(float3(h, real(1.f), real(1.f)));
		color = color / RGBtoBrightness(color);
		output[i].used = true;
		output[i].pos = float4(r_cos(phi) * real(5.f), real(3.f), r_sin(phi) * real(5.f), real(0.f));
		output[i].extend = real(0.25f);
		output[i].falloff = real(0.25f);
		output[i].color = color * real(0.5f);
	}
}

template <class Scene>
inline real map_geometry(const Frame &F, const GeometryInput &geometry, const MarchingInput &march)
{
	float3 debug_plane_point = get_debug_plane_point(F);
	float3 debug_plane_normal = get_debug_plane_normal(F);

	real output_scene_distance = 3e38f;

	MaterialInput material_input;
	material_input.obj_normal = float3(real(0.f));
	material_input.iteration_count = 0;
	material_input.scene_distance = 0.f;
	MaterialOutput material_output = zero_material_output();

	if (debug_show_objects(F))
	{
		Scene::map(F, geometry, march, material_input, material_output, true, output_scene_distance);
	}
	real distance_debug_plane = sdPlaneFast(geometry.pos - debug_plane_point, geometry.dir, debug_plane_normal);

	if (any(debug_plane_normal))
		return r_min(output_scene_distance, distance_debug_plane);
	else
		return output_scene_distance;
}

// :137-162
template <class Scene>
inline void map_material(const Frame &F, GeometryInput geometry, const MaterialInput &material_input, MaterialOutput &material_output)
{
	float3 debug_plane_point = get_debug_plane_point(F);
	float3 debug_plane_normal = get_debug_plane_normal(F);
	real debug_plane_scale = F.debug_scale;

	real output_scene_distance = 3e38f;
	MarchingInput march;
	march.is_inside = false;
	march.last_transparent_pos = float3(real(0.f));
	march.has_transparent = false;
	march.is_shadow_pass = false;
	real distance_debug_plane = sdPlaneFast(geometry.pos - debug_plane_point, geometry.dir, debug_plane_normal);
	if (any(debug_plane_normal) && material_hit(distance_debug_plane))
	{
		MaterialInput material_input_dummy;
		material_input_dummy.obj_normal = float3(real(0.f));
		material_input_dummy.iteration_count = 0;
		material_input_dummy.scene_distance = 0.f;
		MaterialOutput material_output_dummy = zero_material_output();

		geometry.dir.w = 0.f;
		Scene::map(F, geometry, march, material_input_dummy, material_output_dummy, true, output_scene_distance);

		material_output.material_id = MATERIAL_DISTANCE_PLANE;
		material_output.material_properties.x = output_scene_distance / debug_plane_scale;
	}
	else
	{
		Scene::map(F, geometry, march, material_input, material_output, false, output_scene_distance);
	}
}

// :164-177
template <class Scene>
inline float3 grad(const Frame &F, GeometryInput geometry, const MarchingInput &march, real baseline, real sample_distance)
{
	float3 pos = geometry.pos;

	geometry.pos = pos + float3(sample_distance, real(0.f), real(0.f));
	real d1 = map_geometry<Scene>(F, geometry, march) - baseline;
	geometry.pos = pos + float3(real(0.f), sample_distance, real(0.f));
	real d2 = map_geometry<Scene>(F, geometry, march) - baseline;
	geometry.pos = pos + float3(real(0.f), real(0.f), sample_distance);
	real d3 = map_geometry<Scene>(F, geometry, march) - baseline;

	return normalize(float3(d1, d2, d3));
}

// :179-220.  `iter` keeps its value after break/return and `continue` still increments
// it (a-T.10).
template <class Scene>
inline bool march_ray(const Frame &F, GeometryInput &geometry, const MarchingInput &march, real dist_max, real inside_sign,
	uint &iter, real &scene_distance, PixelStats &stats)
{
	float3 start_pos = geometry.pos;
	geometry.camera_distance = 0.f;
	real step_factor = 1.0f;
	real last_scene_distance = 0.f;
	real last_safe_camera_distance = 0.f;
	scene_distance = 0.f;
	for (iter = 0; iter < (uint)F.iter_count; ++iter)
	{
		if (iter == 3)
		{
			step_factor = 1.5f;
		}

		geometry.pos = mad(geometry.dir.xyz(), geometry.camera_distance, start_pos);
		scene_distance = map_geometry<Scene>(F, geometry, march) * inside_sign;
		stats.march_evals++;
		// check for overstepping
		if (step_factor > real(1.f) && (last_scene_distance + scene_distance) < last_scene_distance * step_factor)
		{
			geometry.camera_distance = last_safe_camera_distance;
			step_factor = 1.f;
			continue;
		}
		last_scene_distance = scene_distance;

		if (geometry.camera_distance > dist_max)
		{
			return false;
		}
		else if (scene_distance < real(dist_eps))
		{
			return true;
		}

		last_safe_camera_distance = geometry.camera_distance + scene_distance;
		geometry.camera_distance = geometry.camera_distance + scene_distance * step_factor;
	}
	return false;
}

// :222-233
inline uint find_next_ray(const Ray *rays, int ray_slots)
{
	uint ray_index = 0;
	for (uint index = 1; index < (uint)ray_slots; ++index)
	{
		if (rays[index].depth < rays[ray_index].depth)
			ray_index = index;
	}
	return ray_index;
}
// :235-246
inline uint find_free_ray(const Ray *rays, int ray_slots)
{
	uint index;
	for (index = 0; index < (uint)ray_slots; ++index)
	{
		if (rays[index].depth == INVALID_DEPTH)
			break;
	}
	return index;
}

// :260-639, for the pixel (px, py) of a width x height target; row 0 is the top row.
template <class Scene>
inline float4 ps_main(const Frame &F, int px, int py, PixelStats &stats)
{
	const int RAY_COUNT = F.ray_count;
	const int LIGHT_COUNT = F.light_count;
	const real RANGE = F.range;

	// pixel centre -> NDC (vshader.hlsl:12-16 + rasteriser; a1)
	real screen_x = (real((float)px) + real(0.5f)) / real((float)F.width) * real(2.f) - real(1.f);
	real screen_y = real(1.f) - (real((float)py) + real(0.5f)) / real((float)F.height) * real(2.f);
	// ddx/ddy of the linear interpolant are constants (a-T.9)
	real ddx_x = real(2.f) / real((float)F.width);
	real ddy_y = real(-2.f) / real((float)F.height);

	// main ray (:263-267)
	float3 dir = F.front_vec + screen_x * F.right_vec + screen_y * F.top_vec;
	real dir_invlen = real(1.f) / length(dir);
	dir = dir * dir_invlen;
	float3 right_ray_vec = ddx_x * F.right_vec * dir_invlen;
	float3 bottom_ray_vec = ddy_y * F.top_vec * dir_invlen;

	Ray rays[MAX_RAY_COUNT];
	for (int index = 0; index < MAX_RAY_COUNT; ++index)
	{
		// payload of slots 1..7 is uninitialised in the reference (Q11); zero it here
		rays[index].pos = rays[index].dir = rays[index].contribution = rays[index].last_transparent_pos = float3(real(0.f));
		rays[index].inside_sign = 0.f;
		rays[index].has_transparent = false;
		rays[index].shadow_range = 0.f;
		rays[index].is_shadow_ray = false;
		rays[index].depth = INVALID_DEPTH;
	}

	rays[0].pos = F.eye;
	rays[0].dir = dir;
	rays[0].contribution = float3(real(1.f), real(1.f), real(1.f));
	rays[0].inside_sign = 1.f;
	rays[0].last_transparent_pos = float3(real(0.f));
	rays[0].has_transparent = false;
	rays[0].shadow_range = 0.f;
	rays[0].is_shadow_ray = false;
	rays[0].depth = 0;
	uint ray_count = 1;

	real hdr_output = -1.f;
	float4 out_color = float4(real(0.f), real(0.f), real(0.f), real(0.f));
	for (uint bounce = 0; bounce < (uint)F.bounce_count && ray_count > 0; ++bounce)
	{
		stats.rays++;
		// get next ray (:289-294)
		uint ray_index = find_next_ray(rays, RAY_COUNT);
		Ray current_ray = rays[ray_index];
		rays[ray_index].depth = INVALID_DEPTH;
		--ray_count;

		float3 output_color = float3(real(0.f));
		// march geometry (:299-316)
		GeometryInput geometry_input;
		geometry_input.pos = current_ray.pos;
		geometry_input.dir = float4(current_ray.dir, real(1.f));
		geometry_input.camera_distance = 0.f;
		geometry_input.right_ray_offset = right_ray_vec;
		geometry_input.bottom_ray_offset = bottom_ray_vec;

		MarchingInput marching_input;
		marching_input.is_inside = false;
		marching_input.has_transparent = current_ray.has_transparent;
		marching_input.last_transparent_pos = current_ray.last_transparent_pos;
		marching_input.is_shadow_pass = current_ray.is_shadow_ray;

		real max_range = current_ray.is_shadow_ray ? current_ray.shadow_range : RANGE;

		uint iter_count;
		real scene_distance;
		bool scene_hit = march_ray<Scene>(F, geometry_input, marching_input, max_range, current_ray.inside_sign, iter_count, scene_distance, stats);
		if (scene_hit)
		{
			stats.hits++;
			// normal, first pass (:320-330)
			NormalOutput normal_output;
			normal_output.use_normal = false;
			normal_output.normal = float3(real(0.f));
			normal_output.normal_sample_dist = grad_eps;

			geometry_input.dir.w = 0.f;
			Scene::map_normal(F, geometry_input, normal_output);
			if (!normal_output.use_normal)
			{
				normal_output.normal = grad<Scene>(F, geometry_input, marching_input, scene_distance * current_ray.inside_sign, normal_output.normal_sample_dist);
			}

			// material (:333-353)
			MaterialInput material_input;
			material_input.obj_normal = normal_output.normal;
			material_input.iteration_count = iter_count;
			material_input.scene_distance = scene_distance;

			MaterialOutput material_output;
			material_output.material_id = MATERIAL_NONE;
			material_output.material_position = float4(geometry_input.pos, real(0.f));
			material_output.material_properties = float4(real(0.f));
			material_output.diffuse_color = float4(real(0.f), real(0.f), real(0.f), real(1.f));
			material_output.specular_color = float4(real(0.f), real(0.f), real(0.f), real(60.f));
			material_output.emissive_color = float3(real(0.f));
			material_output.reflection_color = float3(real(0.f));
			material_output.refraction_color = float3(real(0.f));
			material_output.optical_index = 1.4f;
			material_output.optical_density = 0.f;
			material_output.normal = float4(real(0.f));
			material_output.max_cost = F.max_cost_default;
			material_output.use_hdr = true;

			map_material<Scene>(F, geometry_input, material_input, material_output);
			// EXTENSION, off by default (BASELINE configs[2] as worded: "2 reflection bounces"; the reference's
			// labyrinth has no reflective material): marble is given a reflection colour
			if (F.extension_marble_reflection != real(0.f) &&
				(material_output.material_id == MATERIAL_MARBLE_DARK || material_output.material_id == MATERIAL_MARBLE_LIGHT))
				material_output.reflection_color = float3(F.extension_marble_reflection);

			if (!current_ray.is_shadow_ray)
			{
				// hdr flag (:357-359, Q2)
				real new_hdr = material_output.use_hdr ? real(1.f) : real(0.f);
				hdr_output = r_lerp(hdr_output, new_hdr, r_step(hdr_output, real(0.f)));

				// normal, second pass (:362)
				float3 new_normal = lerp(normal_output.normal, material_output.normal.xyz(), material_output.normal.w);

				// reflection (:365-384)
				if (any(material_output.reflection_color) && current_ray.inside_sign > real(0.f) && current_ray.depth + 3 < material_output.max_cost)
				{
					if (ray_count < (uint)RAY_COUNT)
					{
						uint new_ray_index = find_free_ray(rays, RAY_COUNT);
						float3 ref_vec = reflect(geometry_input.dir.xyz(), new_normal);

						Ray &r = rays[new_ray_index];
						r.pos = mad(ref_vec, real(reflect_eps), geometry_input.pos);
						r.dir = ref_vec;
						r.contribution = material_output.reflection_color * current_ray.contribution;
						r.inside_sign = 1.f;
						r.last_transparent_pos = float3(real(0.f));
						r.has_transparent = false;
						r.shadow_range = 0.f;
						r.is_shadow_ray = false;
						r.depth = current_ray.depth + 3;
						++ray_count;
					}
				}

				// refraction (:387-423)
				if (any(material_output.refraction_color) && current_ray.depth + 4 < material_output.max_cost)
				{
					if (ray_count < (uint)RAY_COUNT)
					{
						uint new_ray_index = find_free_ray(rays, RAY_COUNT);
						Ray &r = rays[new_ray_index];
						if (current_ray.inside_sign > real(0.f)) // entering the material
						{
							float3 ref_vec = refract(geometry_input.dir.xyz(), new_normal, real(1.f) / material_output.optical_index);
							r.pos = mad(ref_vec, real(refract_eps), geometry_input.pos);
							r.dir = ref_vec;
							r.contribution = material_output.refraction_color * current_ray.contribution;
							r.inside_sign = -1.f;
						}
						else // leaving the material
						{
							float3 ref_vec = refract(geometry_input.dir.xyz(), -new_normal, material_output.optical_index);
							r.pos = mad(ref_vec, real(refract_eps), geometry_input.pos);
							r.dir = ref_vec;
							r.contribution = material_output.refraction_color * current_ray.contribution;
							r.inside_sign = 1.f;
						}
						r.last_transparent_pos = float3(real(0.f));
						r.has_transparent = false;
						r.shadow_range = 0.f;
						r.is_shadow_ray = false;
						r.depth = current_ray.depth + 2;
						++ray_count;
					}
				}

				float3 diffuse_color = material_output.diffuse_color.xyz();
				float3 color = float3(real(0.f));

				bool use_light = true;
				// material switch (:430-481)
				if (material_output.material_id == MATERIAL_ITER)
				{
					color = color + iter_count_to_color(iter_count, (uint)(F.iter_count - 1));
					use_light = false;
					hdr_output = 0.f;
				}
				else if (material_output.material_id == MATERIAL_PLAIN)
				{
					color = color + diffuse_color;
					use_light = false;
				}
				else if (material_output.material_id == MATERIAL_NORMAL1)
				{
					float3 normal_color = v_max(real(0.01f), new_normal);
					normal_color = normal_color / r_max(r_max(normal_color.x, normal_color.y), normal_color.z);
					color = color + normal_color;
					use_light = false;
					hdr_output = 0.f;
				}
				else if (material_output.material_id == MATERIAL_NORMAL2)
				{
					color = color + v_abs(new_normal);
					use_light = false;
					hdr_output = 0.f;
				}
				else if (material_output.material_id == MATERIAL_DISTANCE_PLANE)
				{
					color = color + debug_plane_color(material_output.material_properties.x);
					use_light = false;
					hdr_output = 0.f;
				}
				else if (material_output.material_id == MATERIAL_WOOD)
				{
					diffuse_color = diffuse_color + wood(material_output.material_position.xyz());
				}
				else if (material_output.material_id == MATERIAL_MARBLE_DARK)
				{
					diffuse_color = diffuse_color + marble(material_output.material_position.xyz(), float3(real(0.556f), real(0.478f), real(0.541f)));
				}
				else if (material_output.material_id == MATERIAL_MARBLE_LIGHT)
				{
					diffuse_color = diffuse_color + marble(material_output.material_position.xyz(), float3(real(0.7f), real(0.7f), real(0.7f)));
				}
				else if (material_output.material_id == MATERIAL_FIRE)
				{
					real fadeout = r_saturate(dot(-geometry_input.dir.xyz(), new_normal));
					float4 fire_color = fire(material_output.material_position.xyz(), real(1.f) - fadeout);
					color = color + fire_color.xyz();
					material_output.diffuse_color.w = r_saturate(fire_color.w);
					material_output.diffuse_color.x = material_output.diffuse_color.y = material_output.diffuse_color.z = real(1.f);
				}

				// transparent material (:484-501, Q8)
				if (material_output.diffuse_color.w < real(1.f) && current_ray.depth + 2 < material_output.max_cost)
				{
					if (ray_count < (uint)RAY_COUNT)
					{
						uint new_ray_index = find_free_ray(rays, RAY_COUNT);
						Ray &r = rays[new_ray_index];
						r.pos = geometry_input.pos;
						r.dir = geometry_input.dir.xyz();
						r.contribution = (real(1.f) - material_output.diffuse_color.w) * material_output.diffuse_color.xyz() * current_ray.contribution;
						r.inside_sign = 1.f;
						r.last_transparent_pos = geometry_input.pos;
						r.has_transparent = true;
						r.shadow_range = 0.f;
						r.is_shadow_ray = false;
						r.depth = current_ray.depth + 2;
						++ray_count;
					}
				}

				if (use_light)
				{
					// :505-516
					LightOutput light_output[MAX_LIGHT_COUNT];
					for (int i1 = 0; i1 < MAX_LIGHT_COUNT; ++i1)
					{
						light_output[i1].used = false;
						light_output[i1].pos = float4(real(0.f));
						light_output[i1].color = float3(real(0.f));
						light_output[i1].falloff = 0.f;
						light_output[i1].extend = 0.f;
					}

					real ambient_lighting_factor = 0.075f;
					Scene::map_light(F, geometry_input, light_output, ambient_lighting_factor);
					extension_lights(F, light_output);

					// :519-521
					float3 view_dir = geometry_input.dir.xyz();
					real shadow_move_distance = r_max(real(shadow_eps), normal_output.normal_sample_dist) + r_max(real(0.f), -scene_distance);
					float3 scene_pos = mad(new_normal, shadow_move_distance, geometry_input.pos);

					// :524-587
					for (int i2 = 0; i2 < LIGHT_COUNT; ++i2)
					{
						if (light_output[i2].used)
						{
							float3 lighting_dir;
							real distance_to_trace;
							real falloff_factor = 1.f;
							if (light_output[i2].pos.w == real(1.f)) // directional light
							{
								lighting_dir = light_output[i2].pos.xyz();
								lighting_dir = lighting_dir / (length(lighting_dir) + real(dist_eps));
								distance_to_trace = RANGE;
							}
							else // point light (Q3: distance-independent falloff)
							{
								lighting_dir = scene_pos - light_output[i2].pos.xyz();
								distance_to_trace = length(lighting_dir);
								lighting_dir = lighting_dir / distance_to_trace;
								distance_to_trace -= light_output[i2].extend;

								falloff_factor = r_pow(real(0.1f), light_output[i2].falloff);
							}
							float3 light_color = light_output[i2].color * falloff_factor;

							// ambient (:550)
							color = color + diffuse_color * light_color * ambient_lighting_factor;

							float3 light_influenced_color = float3(real(0.f));

							// diffuse (:557-558)
							real light_dot = r_saturate(dot(-new_normal, lighting_dir));
							light_influenced_color = light_influenced_color + diffuse_color * light_color * light_dot;

							// specular (:561-565)
							float3 half_vec = -normalize(view_dir + lighting_dir);
							real specular_dot = r_saturate(dot(new_normal, half_vec));
							real specular_factor = r_pow(specular_dot, material_output.specular_color.w);

							light_influenced_color = light_influenced_color + material_output.specular_color.xyz() * light_color * specular_factor;

							// shadow ray (:568-585, F6)
							if (current_ray.depth + 2 < material_output.max_cost && light_dot > real(0.f))
							{
								if (ray_count < (uint)RAY_COUNT)
								{
									uint new_ray_index = find_free_ray(rays, RAY_COUNT);
									Ray &r = rays[new_ray_index];
									r.pos = scene_pos;
									r.dir = -lighting_dir;
									r.contribution = light_influenced_color * current_ray.contribution * r_saturate(material_output.diffuse_color.w);
									r.inside_sign = 1.f;
									r.last_transparent_pos = float3(real(0.f));
									r.has_transparent = false;
									r.shadow_range = distance_to_trace;
									r.is_shadow_ray = true;
									r.depth = current_ray.depth + 2;
									++ray_count;
								}
							}
						}
					}

					// emissive + alpha (:590-593)
					color = color + material_output.emissive_color;
					color = color * r_saturate(material_output.diffuse_color.w);
				}

				output_color = output_color + color * current_ray.contribution;
			}
			else // shadow ray hit something (:598-619)
			{
				if (material_output.diffuse_color.w < real(1.f) && current_ray.depth + 2 < material_output.max_cost)
				{
					if (ray_count < (uint)RAY_COUNT)
					{
						uint new_ray_index = find_free_ray(rays, RAY_COUNT);
						Ray &r = rays[new_ray_index];
						r.pos = geometry_input.pos;
						r.dir = geometry_input.dir.xyz();
						r.inside_sign = 1.f;
						r.last_transparent_pos = geometry_input.pos;
						r.has_transparent = true;
						r.contribution = (real(1.f) - material_output.diffuse_color.w) * material_output.diffuse_color.xyz() * current_ray.contribution;
						r.shadow_range = max_range - geometry_input.camera_distance;
						r.is_shadow_ray = true;
						r.depth = current_ray.depth + 2;
						++ray_count;
					}
				}
			}
		}
		else // scene not hit (:621-632)
		{
			if (current_ray.is_shadow_ray)
			{
				output_color = output_color + current_ray.contribution;
			}
			else
			{
				float3 background_color = Scene::map_background(F, geometry_input.dir.xyz(), iter_count);
				output_color = output_color + background_color * current_ray.contribution;
			}
		}
		out_color.x += output_color.x;
		out_color.y += output_color.y;
		out_color.z += output_color.z;
	}

	// :636-638
	out_color.w = r_abs(hdr_output);
	return out_color;
}

} // namespace orc
