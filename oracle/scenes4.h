// oracle/scenes4.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates Engine/shader/scenes/sdf_scene_tree.hlsl, the last of the
// reference's 22 scene plugins.
#pragma once
#include "scenes3.h"

namespace orc {

struct SceneTree
{
	// :5-17 (five-argument overload; unused by the scene, kept for completeness of the restatement)
	static real sdBranch5(float3 pos, real h1, real h2, real h3, real r1, real r2)
	{
		float2 p2 = float2(length(float2(pos.x, pos.z)), pos.y);
		real plane1 = dot(p2, normalize(float2(h1, -r1)));
		real plane2 = dot(p2 - float2(r1, h1), normalize(float2(h2 - h1, r1 - r2)));
		real plane3 = dot(p2 - float2(real(0.f), h3), normalize(float2(h3 - h2, r2)));
		real plane_bottom = -pos.y;
		real plane_top = pos.y - h3;
		return r_max(r_max(r_max(r_max(plane1, plane2), plane3), plane_bottom), plane_top);
	}
	// :19-28
	static real sdBranch(float3 pos, real h2, real r1, real r2)
	{
		float2 p2 = float2(length(float2(pos.x, pos.z)), pos.y);
		real plane = dot(p2 - float2(r1, real(0.f)), normalize(float2(h2, r1 - r2)));
		real plane_bottom = -pos.y;
		real plane_top = pos.y - h2;
		return r_max(r_max(plane, plane_bottom), plane_top);
	}
	// :30-76
	static void sdTree(float3 pos, real &tree, real &leafes, real noiseval)
	{
		real tree_scale = 1.f;
		real leaf_scale = 1.f;
		tree = 3e30f;
		leafes = 3e30f;

		real angle1 = 35.f;
		real angle2 = 34.f;
		real angle3 = real(90.f) - noiseval * real(10.f);
		real side_offset = 0.075f;
		real height_offset1 = real(0.33f) + noiseval * real(0.05f);
		real height_offset2 = 0.41f;
		real sphere_size = real(0.07f) - noiseval * real(0.02f);
		const uint iters = 9;
		real tree_scale_factor = 1.4f;
		real leaf_scale_factor = 1.3f;

		for (uint i = 0; i < iters; ++i)
		{
			// same quotients as pos / tree_scale; r_div_const additionally records, in the census build,
			// numerators outside the domain on which the kernels' constant division is proven exact
			const float3 scaled_pos = float3(r_div_const(pos.x, tree_scale), r_div_const(pos.y, tree_scale), r_div_const(pos.z, tree_scale));
			real branch = sdBranch(scaled_pos, real(1.f), real(0.1f), real(0.05f)) * tree_scale;
			tree = smin(tree, branch, real(0.01f));

			real leaf = sdSphere(scaled_pos - float3(real(0.f), real(1.f) + sphere_size * leaf_scale, real(0.f)), sphere_size * leaf_scale) * tree_scale;
			leafes = r_min(leafes, leaf);

			real height = (i == 0) ? height_offset1 : height_offset2;
			pos.y -= height * tree_scale;
			pos.x = r_abs(pos.x);
			pos.z = r_abs(pos.z);
			if (pos.x > pos.z && i == 0)
			{
				real t = pos.x;
				pos.x = pos.z;
				pos.z = t;
			}
			pos.z += side_offset * tree_scale;
			real angle = (i == 0) ? angle1 : angle2;
			float2 yz = opRotate(float2(pos.y, pos.z), -angle / real(180.f) * real(pi));
			pos.y = yz.x;
			pos.z = yz.y;
			float2 xz = opRotate(float2(pos.x, pos.z), angle3 / real(180.f) * real(pi));
			pos.x = xz.x;
			pos.z = xz.y;

			tree_scale /= tree_scale_factor;
			leaf_scale *= leaf_scale_factor;
		}
	}
	// :80-126: voronoi with the distance to the cell border ALONG a direction
	static void voronoi_dir(float2 uv, float2 dir, real max_offset, float2 &closest_cell_id, float2 &closest_center_vec, real &closest_distance)
	{
		float2 cell_index = v_floor(uv);
		float2 cell_pos = (uv - cell_index) - real(0.5f);
		real l_center_min = 10.f;
		closest_distance = 10.f;
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
					closest_center_vec = center_vec;
				}
			}
		for (int x = -1; x < 2; ++x)
			for (int y = -1; y < 2; ++y)
			{
				float2 offset = float2(real((float)x), real((float)y));
				float2 cell_id = cell_index + offset;
				float2 point_pos = offset + voronoi_cell_offset(cell_id) * max_offset;
				float2 border_vec = (point_pos + closest_cell) * real(0.5f);
				float2 normal_vec = normalize(closest_cell - border_vec);
				real edge_dist = r_abs(dot(normal_vec, cell_pos - border_vec));
				real dir_distance = edge_dist / r_max(dot(normal_vec, -dir), real(0.0001f));
				closest_distance = r_min(closest_distance, dir_distance);
			}
	}
	// :128-143
	static float2 jump(real slide_time, real jump_time, real stime)
	{
		real total_time = slide_time + jump_time;
		real cycle_pos = stime - r_floor(stime / total_time) * total_time;
		if (cycle_pos < slide_time)
		{
			return float2(cycle_pos / slide_time, real(0.f));
		}
		else
		{
			real jump_cycle = (cycle_pos - slide_time) / jump_time;
			real x = real(1.f) - jump_cycle;
			real y = real(4.f) * (jump_cycle - jump_cycle * jump_cycle);
			return float2(x, y);
		}
	}
	// :145-228
	static void map(const Frame &F, const GeometryInput &geometry, const MarchingInput &march, const MaterialInput &material_input,
		MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
	{
		real bounding = sdPlaneFast(geometry.pos - float3(real(0.f), real(2.f), real(0.f)), geometry.dir, float3(real(0.f), real(1.f), real(0.f)));

		real tree = 1e30f, leafes = 1e30f;
		real eye = 1e30f, pupil = 1e30f;
		float2 cell_index = float2(real(0.f), real(0.f));
		real noise_val = 0.f;

		if (bounding < real(0.1f))
		{
			float3 pos = geometry.pos;
			pos.z -= F.stime / real(10.f) * real(0.4f);

			real tree_distance = 2.2f;
			real closest_distance;
			float2 cell_pos = float2(real(0.f));
			voronoi_dir(float2(pos.x, pos.z) / tree_distance, normalize(float2(geometry.dir.x, geometry.dir.z)), real(0.3f), cell_index, cell_pos, closest_distance);
			noise_val = r_sin(cell_index.x * real(356.12f) + cell_index.y + real(82.6f)) * real(0.5f) + real(0.5f);

			float2 jump_offset = jump(real(10.f), real(1.f), F.stime + noise_val * real(10.f)) / tree_distance;
			cell_pos.y -= jump_offset.x * real(0.4f) - real(0.05f);

			float2 cs = cell_pos * tree_distance;
			float3 tree_pos = float3(cs.x, pos.y - jump_offset.y, cs.y);
			cell_pos = opRotate(cell_pos, noise_val);
			float2 cr = cell_pos * tree_distance;
			float3 tree_pos_rotated = float3(cr.x, pos.y - jump_offset.y, cr.y);

			sdTree(tree_pos_rotated, tree, leafes, noise_val);
			tree_pos.x = r_abs(tree_pos.x);
			eye = sdSphere(tree_pos - float3(real(0.2f), real(1.f), real(-0.5f)), real(0.12f));
			pupil = sdSphere(tree_pos - float3(real(0.2f), real(1.f), real(-0.59f)), real(0.05f));

			tree = r_min(tree, closest_distance * tree_distance + real(0.1f));
			leafes = r_min(leafes, closest_distance * tree_distance + real(0.1f));
		}

		real ground_plane = sdPlaneFast(geometry.pos, geometry.dir, float3(real(0.f), real(1.f), real(0.f)));

		if (geometry_step)
		{
			if (bounding >= real(0.1f))
			{
				object_add(output_scene_distance, bounding);
			}
			object_add(output_scene_distance, tree);
			object_add(output_scene_distance, leafes);
			object_add(output_scene_distance, eye);
			object_add(output_scene_distance, pupil);
			object_add(output_scene_distance, ground_plane);
		}
		else
		{
			if (material_hit(tree))
			{
				material_output.diffuse_color.x = real(0.5f);
				material_output.diffuse_color.y = real(0.25f);
				material_output.diffuse_color.z = real(0.1f);
				set_rgb(material_output.specular_color, real(0.15f));
			}
			else if (material_hit(leafes))
			{
				float3 green = lerp(float3(real(0.2f), real(0.9f), real(0.2f)), float3(real(0.3f), real(0.5f), real(0.2f)), noise_val);
				material_output.diffuse_color.x = green.x;
				material_output.diffuse_color.y = green.y;
				material_output.diffuse_color.z = green.z;
				set_rgb(material_output.specular_color, real(0.15f));
			}
			else if (material_hit(eye))
			{
				set_rgb(material_output.diffuse_color, real(0.9f));
				set_rgb(material_output.specular_color, real(0.15f));
			}
			else if (material_hit(pupil))
			{
				set_rgb(material_output.diffuse_color, real(0.1f));
				set_rgb(material_output.specular_color, real(0.15f));
			}
			else if (material_hit(ground_plane))
			{
				real turb = turbulence(geometry.pos);
				float3 brown1 = float3(real(218.f), real(173.f), real(136.f)) / real(255.f);
				float3 brown2 = float3(real(140.f), real(90.f), real(60.f)) / real(255.f);
				float3 brown = lerp(brown1, brown2, turb);
				float3 c = brown * real(0.6f);
				material_output.diffuse_color.x = c.x;
				material_output.diffuse_color.y = c.y;
				material_output.diffuse_color.z = c.z;
				set_rgb(material_output.specular_color, real(0.05f));
			}
		}
	}
	static void map_normal(const Frame &, const GeometryInput &, NormalOutput &) {}
	// :234-241
	static void map_light(const Frame &, const GeometryInput &, LightOutput *output, real &ambient_lighting_factor)
	{
		output[0].used = true;
		output[0].pos = float4(real(-1.f), real(-1.f), real(1.2f), real(1.f));
		output[0].color = float3(real(1.f), real(1.f), real(1.f)) * real(1.3f);
		ambient_lighting_factor = 0.2f;
	}
	static float3 map_background(const Frame &F, float3 dir, uint) { return sky_color(dir, F.stime); }
};

} // namespace orc
