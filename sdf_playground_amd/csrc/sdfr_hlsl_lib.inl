// sdfr_hlsl_lib.inl -- the reference's shader libraries under their HLSL names, as MEMBERS of a scene class written in the
// reference's dialect (included inside the class body by the translation unit sdfr_hlsl.cpp generates, so that a scene's own
// functions overload them, as they do in HLSL, instead of hiding them).  Each is a thin wrapper of the function the built-in
// scenes use (sdfr_lib.h, sdfr_noise.h): same arithmetic, same bits.
//   sdf_primitives.hlsl:6-131   sdf_ops.hlsl:6-134   sdf_common.hlsl:4-94   sdf_materials.hlsl:6-201   noise.hlsl:6-16,76-476

// ---- primitives ----
SDF_HD float sdSphere(float3 pos, float radius) const { return sd_sphere(pos, radius); }
SDF_HD float sdSphereFast(float3 pos, float4 dir, float r) const { return sd_sphere_fast(pos, V3(dir.x, dir.y, dir.z), any(dir.w), r, dist_eps); }
SDF_HD float sdBox(float3 pos, float3 size) const { return sd_box(pos, size); }
SDF_HD float sdPlane(float3 pos, float3 plane_norm) const { return sd_plane(pos, plane_norm); }
SDF_HD float sdPlaneFast(float3 pos, float4 dir, float3 plane_norm) const { return sd_plane_fast(pos, V3(dir.x, dir.y, dir.z), any(dir.w), plane_norm); }
SDF_HD float sdTorusXY(float3 pos, float radius_big, float radius_small) const { return sd_torus_xy(pos, radius_big, radius_small); }
SDF_HD float sdCappedCylinder(float3 pos, float h, float r) const { return sd_capped_cylinder(pos, h, r); }
SDF_HD float sdRoundCone(float3 p, float3 a, float3 b, float r1, float r2) const { return sd_round_cone(p, a, b, r1, r2); }
SDF_HD float sdLimit1(float pos, float dir, float lim_val) const { return sd_limit1(pos, dir, lim_val); }
SDF_HD float sdLimit2(float2 pos, float2 dir, float2 lim_val) const { return sd_limit2(pos, dir, lim_val); }
SDF_HD float sdLimit3(float3 pos, float3 dir, float3 lim_val) const { return sd_limit3(pos, dir, lim_val); }
// ---- operators ----
SDF_HD float3 opRepLim(float3 pos, float3 count, float3 size) const { return float3(op_rep_lim((vec3)pos, (vec3)count, (vec3)size)); }
SDF_HD float2 opRepLim(float2 pos, float2 count, float2 size) const { return float2(op_rep_lim((vec2)pos, (vec2)count, (vec2)size)); }
SDF_HD float opRepLim(float pos, float count, float size) const { return op_rep_lim(pos, count, size); }
SDF_HD float3 opRepInf(float3 pos, float3 size) const { return float3(op_rep_inf((vec3)pos, (vec3)size)); }
SDF_HD float2 opRepInf(float2 pos, float2 size) const { return float2(op_rep_inf((vec2)pos, (vec2)size)); }
SDF_HD float opRepInf(float pos, float size) const { return op_rep_inf(pos, size); }
SDF_HD float opRepAngle(float2 &pos, float count) const
{
	vec2 p = pos;
	const float index = op_rep_angle(&p, count);
	pos = float2(p);
	return index;
}
// an inout parameter handed a swizzle (`opRepAngle(obj_pos.xz, 8.f)`): copy in, copy out
template <int A, int B>
SDF_HD float opRepAngle(swz2<A, B> &pos, float count) const
{
	float2 p = pos;
	const float index = opRepAngle(p, count);
	pos = p;
	return index;
}
SDF_HD float2 opRotate(float2 pos, float angle) const { return float2(op_rotate(pos, angle)); }
SDF_HD float opShell(float distance, float inner, float outer) const { return op_shell(distance, inner, outer); }
SDF_HD float2 opAB2UV(float2 input) const { return float2(op_ab2uv(input)); }
SDF_HD float opChamfer(float a, float b, float size) const { return op_chamfer(a, b, size); }
SDF_HD float opChamferMerge(float a, float b, float size) const { return op_chamfer_merge(a, b, size); }
SDF_HD float opPipe(float a, float b, float size, float count) const { return op_pipe(a, b, size, count); }
SDF_HD float opPipeMerge(float a, float b, float size, float count) const { return op_pipe_merge(a, b, size, count); }
SDF_HD float staircase(float x, float stepval, float spread) const { return op_staircase(x, stepval, spread); }
SDF_HD float smin(float a, float b, float k) const { return op_smin(a, b, k); }
SDF_HD float smax1(float a, float b, float k) const { return op_smax1(a, b, k); }
SDF_HD float smax2(float a, float b, float k) const { return op_smax2(a, b, k); }
// ---- colours, checker floor, sky ----
SDF_HD float3 HUEtoRGB(float H) const { return float3(hue_to_rgb(H)); }
SDF_HD float3 HSVtoRGB(float3 HSV) const { return float3(hsv_to_rgb(HSV)); }
SDF_HD float RGBtoBrightness(float3 rgb) const { return rgb_to_brightness(rgb); }
SDF_HD float2 get_tile_impact(float3 pos, float3 dir) const
{
	const float to_move = pos.y / dir.y;
	return float2(pos.x, pos.z) - float2(dir.x, dir.z) * to_move;
}
SDF_HD float4 tile_color_from_pos(float2 pos) const
{
	const float2 tile_index = floor(pos);
	const float2 tile_pos = pos - tile_index;
	const float tile_parity = round(frac((tile_index.x + tile_index.y) * 0.5f + 0.25f));
	const float grey = tile_parity > 0.5f ? 0.1f : 0.8f;
	const float2 dist_vec = 0.5f - abs(tile_pos - 0.5f);
	return float4(grey, grey, grey, min(dist_vec.x, dist_vec.y));
}
SDF_HD float3 total_tile_color(float3 pos, float3 dir, float3 offset_right, float3 offset_bottom) const
{
	return float3(checker_color(pos, dir, offset_right, offset_bottom));
}
SDF_HD void map_groundplane(GeometryInput geometry, MaterialOutput &material_output, bool geometry_step, float &output_scene_distance) const
{
	const float floor_distance = sdPlaneFast(geometry.pos, geometry.dir, float3(0.f, 1.f, 0.f));
	if (geometry_step)
	{
		OBJECT(floor_distance);
	}
	else if (MATERIAL(floor_distance))
	{
		const float3 offset_right = geometry.right_ray_offset * geometry.camera_distance;
		const float3 offset_bottom = geometry.bottom_ray_offset * geometry.camera_distance;
		material_output.diffuse_color = float4(total_tile_color(geometry.pos, geometry.dir.xyz, offset_right, offset_bottom), 1.f);
		material_output.specular_color.rgb = 1.f;
	}
}
SDF_HD float3 sky_color(float3 dir, float phase) const
{
	const vec2 sc = sincos1(-phase * 0.025f);
	return float3(sdfr::sky_color(dir, sc.x, sc.y));
}
// ---- materials ----
SDF_HD float3 marble(float3 pos, float3 marble_color) const { return float3(mat_marble(pos, marble_color)); }
SDF_HD float3 wood(float3 pos) const { return float3(mat_wood(pos)); }
SDF_HD float4 fire(float3 pos, float threshold) const { return float4(mat_fire(pos, threshold)); }
SDF_HD float2 voronoi_cell_offset(float2 cell_index) const { return float2(voronoi_site(cell_index)); }
SDF_HD float4 voronoi(float2 uv, float max_offset) const { return float4(sdfr::voronoi(uv, max_offset)); }
SDF_HD float4 truchet_band(float2 uv, float chance, float width, float2 miss_uv) const { return float4(sdfr::truchet_band(uv, chance, width, miss_uv)); }
SDF_HD float4 braid(float2 uv, float width, float run_length, float run_flip, float2 miss_uv) const { return float4(sdfr::braid(uv, width, run_length, run_flip, miss_uv)); }
SDF_HD float3 debug_plane_color(float scene_distance) const { return float3(mat_debug_plane(scene_distance)); }
SDF_HD float3 iter_count_to_color(uint iter_count, uint max_iter_count) const { return float3(mat_iter_heat(iter_count, max_iter_count)); }
SDF_HD float coordinate_material(float3 pos, float3 norm, float width) const { return mat_coordinate_grid(pos, norm, width); }
// ---- noise ----
SDF_HD uint hash(uint input) const { return pcg_hash(input); }
SDF_HD float hashf(uint input) const { return pcg_hashf(input); }
SDF_HD float hashf(int input) const { return pcg_hashf((uint)input); }
SDF_HD float mod289(float x) const { return noise_mod289(x); }
SDF_HD float2 mod289(float2 x) const { return float2(noise_mod289(x.x), noise_mod289(x.y)); }
SDF_HD float3 mod289(float3 x) const { return float3(noise_mod289(x.x), noise_mod289(x.y), noise_mod289(x.z)); }
SDF_HD float4 mod289(float4 x) const { return float4(noise_mod289(x.x), noise_mod289(x.y), noise_mod289(x.z), noise_mod289(x.w)); }
SDF_HD float permute(float x) const { return noise_permute(x); }
SDF_HD float3 permute(float3 x) const { return float3(noise_permute(x.x), noise_permute(x.y), noise_permute(x.z)); }
SDF_HD float4 permute(float4 x) const { return float4(noise_permute(x.x), noise_permute(x.y), noise_permute(x.z), noise_permute(x.w)); }
SDF_HD float4 grad4(float j, float4 ip) const { return float4(noise_grad4(j, ip.x, ip.y, ip.z)); }
SDF_HD float snoise(float2 v) const { return snoise2(v); }
SDF_HD float snoise(float3 v) const { return snoise3(v); }
SDF_HD float snoise(float4 v) const { return snoise4(v); }
SDF_HD float turbulence(float3 pos) const { return turbulence3(pos); }
