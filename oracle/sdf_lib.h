// oracle/sdf_lib.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates the scene-plugin ABI types (Engine/shader/sdf_structs.hlsl:4-130) and the
// shader libraries: sdf_primitives.hlsl, sdf_ops.hlsl, sdf_common.hlsl,
// sdf_materials.hlsl.  Each function cites the lines it follows.
#pragma once
#include "hlsl.h"
#include "noise.h"

namespace orc {

// pshader_sdf.hlsl:31-36.  The reference's five `static const float` epsilons, as variables: orc_render copies them from
// the frame before it starts its threads (sdfr_limits exposes them as run-time values; anything but these defaults is a
// labelled extension).  One render at a time per process -- this is test infrastructure.
inline float dist_eps = 0.0001f;
inline float grad_eps = 0.0001f;
inline float reflect_eps = 0.001f;
inline float refract_eps = 0.001f;
inline float shadow_eps = 0.0003f;

// math_constants.hlsl:4-7
static const float sqrt_half = 0.70710678118654752f;
static const float sqrt_two = 1.41421356237309504f;
static const float pi = 3.14159265358979323f;
static const float tau = 6.28318530717958647f;

// pshader_sdf.hlsl:67-76
enum
{
	MATERIAL_NONE = 0, MATERIAL_PLAIN = 1, MATERIAL_ITER = 2, MATERIAL_NORMAL1 = 3, MATERIAL_NORMAL2 = 4,
	MATERIAL_DISTANCE_PLANE = 5, MATERIAL_WOOD = 20, MATERIAL_MARBLE_DARK = 21, MATERIAL_MARBLE_LIGHT = 22,
	MATERIAL_FIRE = 23
};

enum { MAX_RAY_COUNT = 8, MAX_LIGHT_COUNT = 8, MAX_SCENE_VARS = 8 };

// Per-frame uniforms: cbuffer b0 (pshader_sdf.hlsl:17-26), the VAR_ table b1
// (ShaderUtil.cpp:193-267) and the compile-time limits of pshader_sdf.hlsl:60-64,350
// exposed as run-time values (defaults = reference; other values are extensions).
struct Frame
{
	float3 eye, front_vec, right_vec, top_vec;
	real stime;
	int width, height;
	int iter_count;        // ITER_COUNT 100
	int bounce_count;      // BOUNCE_COUNT 16
	int ray_count;         // RAY_COUNT 8
	int light_count;       // LIGHT_COUNT 8
	real range;            // RANGE 100
	uint max_cost_default; // material_output.max_cost = 7
	// driver variables (pshader_sdf.hlsl:88-108,142)
	real debug_nx, debug_ny, debug_nz, debug_scale, debug_x, debug_y, debug_z, show_objects;
	// scene variables in the scene's declaration order
	real scene_var[MAX_SCENE_VARS];
	// EXTENSION (not in the reference; SURVEY.md 8d cfg 5 "8 lights"): number of orbiting point
	// lights placed in slots 1..n after the scene's map_light; 0 = reference behaviour
	int extension_lights;
	// EXTENSION (not in the reference; SURVEY.md 8d cfg 3 "2 reflection bounces"): the labyrinth's marble
	// (MATERIAL_MARBLE_DARK / _LIGHT) gets this reflection_color; 0 = reference behaviour (no reflective material)
	real extension_marble_reflection;
	// pshader_sdf.hlsl:31-35 (reference values 1e-4, 1e-4, 1e-3, 1e-3, 3e-4; anything else is an extension)
	float dist_eps, grad_eps, reflect_eps, refract_eps, shadow_eps;
};

// sdf_structs.hlsl:4-21
struct GeometryInput
{
	float3 pos;
	float4 dir;
	real camera_distance;
	float3 right_ray_offset;
	float3 bottom_ray_offset;
};
// sdf_structs.hlsl:23-37
struct MarchingInput
{
	bool is_inside;
	float3 last_transparent_pos;
	bool has_transparent;
	bool is_shadow_pass;
};
// sdf_structs.hlsl:39-52
struct NormalOutput
{
	real normal_sample_dist;
	float3 normal;
	bool use_normal;
};
// sdf_structs.hlsl:54-64
struct MaterialInput
{
	float3 obj_normal;
	uint iteration_count;
	real scene_distance;
};
// sdf_structs.hlsl:66-110
struct MaterialOutput
{
	uint material_id;
	float4 material_position;
	float4 material_properties;
	float4 diffuse_color;
	float4 specular_color;
	float3 emissive_color;
	float3 reflection_color;
	float3 refraction_color;
	real optical_index;
	real optical_density;
	float4 normal;
	uint max_cost;
	bool use_hdr;
};
// sdf_structs.hlsl:112-130
struct LightOutput
{
	bool used;
	float4 pos;
	real extend;
	float3 color;
	real falloff;
};

inline MaterialOutput zero_material_output()
{
	MaterialOutput m;
	m.material_id = 0;
	m.material_position = float4(real(0.f));
	m.material_properties = float4(real(0.f));
	m.diffuse_color = float4(real(0.f));
	m.specular_color = float4(real(0.f));
	m.emissive_color = float3(real(0.f));
	m.reflection_color = float3(real(0.f));
	m.refraction_color = float3(real(0.f));
	m.optical_index = 0.f;
	m.optical_density = 0.f;
	m.normal = float4(real(0.f));
	m.max_cost = 0;
	m.use_hdr = false;
	return m;
}

// pshader_sdf.hlsl:79-81 (OBJECT / OBJECT_TRANSPARENT / MATERIAL macros)
inline void object_add(real &osd, real distance) { osd = r_min(osd, distance); }
inline void object_add_transparent(real &osd, const MarchingInput &march, real distance, real distance_transparent)
{
	osd = (march.has_transparent && distance_transparent < real(dist_eps)) ? osd : r_min(osd, distance);
}
inline bool material_hit(real distance) { return r_abs(distance) < real(dist_eps); }

// ---- sdf_primitives.hlsl -----------------------------------------------------------

// :6-9
inline real sdSphere(float3 pos, real radius) { return length(pos) - radius; }

// :11-45
inline real sdSphereFast(float3 pos, float4 dir, real r)
{
	if (any(dir.w))
	{
		real b = -dot(pos, dir.xyz());
		real c = dot(pos, pos) - r * r;
		real discriminant = b * b - c;
		if (discriminant < real(0.f))
		{
			return 1e10f;
		}
		else
		{
			real root = r_sqrt(discriminant);
			real t1 = b - root;
			real t2 = b + root;
			if (t1 < real(-dist_eps))
				return (t2 > real(0.f)) ? t2 : real(1e10f);
			else
				return t1;
		}
	}
	else
	{
		return sdSphere(pos, r);
	}
}

// :47-51
inline real sdBox(float3 pos, float3 size)
{
	float3 q = v_abs(pos) - size;
	return length(v_max(q, real(0.f))) + r_min(r_max(q.x, r_max(q.y, q.z)), real(0.f));
}
inline real sdBox(float3 pos, real size) { return sdBox(pos, float3(size)); }

// :53-56
inline real sdPlane(float3 pos, float3 plane_norm) { return dot(pos, plane_norm); }

// :59-70
inline real sdPlaneFast(float3 pos, float4 dir, float3 plane_norm)
{
	real plane_dist = dot(pos, plane_norm);
	if (any(dir.w))
	{
		real denom = r_saturate(dot(dir.xyz(), -plane_norm)) + real(1e-20f);
		ORC_CHECK_PLANE(val(plane_dist), val(denom));
		return plane_dist / denom;
	}
	else
		return plane_dist;
}

// :72-76
inline real sdTorusXY(float3 pos, real radius_big, real radius_small)
{
	float2 q = float2(length(float2(pos.x, pos.y)) - radius_big, pos.z);
	return length(q) - radius_small;
}

// :78-82
inline real sdCappedCylinder(float3 pos, real h, real r)
{
	float2 d = v_abs(float2(length(float2(pos.x, pos.z)), pos.y)) - float2(r, h);
	return r_min(r_max(d.x, d.y), real(0.f)) + length(v_max(d, real(0.f)));
}

// :84-107
inline real sdRoundCone(float3 p, float3 a, float3 b, real r1, real r2)
{
	float3 ba = b - a;
	real l2 = dot(ba, ba);
	real rr = r1 - r2;
	real a2 = l2 - rr * rr;
	real il2 = real(1.0f) / l2;

	float3 pa = p - a;
	real y = dot(pa, ba);
	real z = y - l2;
	float3 x2_s = pa * l2 - ba * y;
	real x2 = dot(x2_s, x2_s);
	real y2 = y * y * l2;
	real z2 = z * z * l2;

	real k = r_sign(rr) * rr * rr * x2;
	if (r_sign(z) * a2 * z2 > k) return r_sqrt(x2 + z2) * il2 - r2;
	if (r_sign(y) * a2 * y2 < k) return r_sqrt(x2 + y2) * il2 - r1;
	return (r_sqrt(x2 * a2 * il2) + y * rr) * il2 - r1;
}

// :110-116
inline real sdLimit1(real pos, real dir, real lim_val)
{
	real barrier_to_use = r_step(real(0.f), dir) - real(0.5f);
	real barrier_pos = barrier_to_use * lim_val - pos;
	return barrier_pos / dir;
}
// :118-124
inline real sdLimit2(float2 pos, float2 dir, float2 lim_val)
{
	float2 barrier_to_use = v_step(real(0.f), dir) - real(0.5f);
	float2 barrier_pos = barrier_to_use * lim_val - pos;
	ORC_CHECK_RAYDIV(val(barrier_pos.x), val(dir.x));
	ORC_CHECK_RAYDIV(val(barrier_pos.y), val(dir.y));
	float2 t = barrier_pos / dir;
	return r_min(t.x, t.y);
}
// :126-132
inline real sdLimit3(float3 pos, float3 dir, float3 lim_val)
{
	float3 barrier_to_use = v_step(real(0.f), dir) - real(0.5f);
	float3 barrier_pos = barrier_to_use * lim_val - pos;
	float3 t = barrier_pos / dir;
	return r_min(r_min(t.x, t.y), t.z);
}

// ---- sdf_ops.hlsl ------------------------------------------------------------------

// :6-25
inline float3 opRepLim(float3 pos, float3 count, float3 size)
{
	float3 rounded = size * (v_round(pos / size + count / real(2.f)) - count / real(2.f));
	float3 limit = count * size * real(0.5f);
	return pos - v_clamp(rounded, -limit, limit);
}
inline float2 opRepLim(float2 pos, float2 count, float2 size)
{
	float2 rounded = size * (v_round(pos / size + count / real(2.f)) - count / real(2.f));
	float2 limit = count * size * real(0.5f);
	return pos - v_clamp(rounded, -limit, limit);
}
inline real opRepLim(real pos, real count, real size)
{
	real rounded = size * (r_round(pos / size + count / real(2.f)) - count / real(2.f));
	real limit = count * size * real(0.5f);
	return pos - r_clamp(rounded, -limit, limit);
}

// :27-43
inline float3 opRepInf(float3 pos, float3 size)
{
	float3 x = pos + size * real(0.5f);
	return x - size * v_floor(x / size) - size * real(0.5f);
}
inline float2 opRepInf(float2 pos, float2 size)
{
	float2 x = pos + size * real(0.5f);
	return x - size * v_floor(float2(r_div_const(x.x, size.x), r_div_const(x.y, size.y))) - size * real(0.5f);
}
inline real opRepInf(real pos, real size)
{
	real x = pos + size * real(0.5f);
	return x - size * r_floor(x / size) - size * real(0.5f);
}

// :45-56 (pos is inout)
inline real opRepAngle(float2 &pos, real count)
{
	real angle = r_atan2(pos.y, pos.x);
	real reduced_angle = r_div_const(angle * count, real(tau)) + real(0.5f);
	real index = r_floor(reduced_angle);
	reduced_angle -= index;
	angle = (reduced_angle - real(0.5f)) * real(tau) / count;
	pos = float2(r_cos(angle), r_sin(angle)) * length(pos);
	return index;
}

// :58-63
inline float2 opRotate(float2 pos, real angle)
{
	real s = r_sin(angle);
	real c = r_cos(angle);
	return float2(pos.x * c - pos.y * s, pos.x * s + pos.y * c);
}

// :68-73
inline real opShell(real distance, real inner, real outer)
{
	real avg = (outer + inner) * real(0.5f);
	real diff = (outer - inner) * real(0.5f);
	return r_abs(distance - avg) - diff;
}

// :77-80
inline float2 opAB2UV(float2 input) { return float2(input.x + input.y, input.x - input.y) * real(sqrt_half); }
// :82-90
inline real opChamfer(real a, real b, real size) { return (a + b - size) * real(sqrt_half); }
inline real opChamferMerge(real a, real b, real size) { return r_min(r_min(a, b), opChamfer(a, b, size)); }

// :92-103
inline real opPipe(real a, real b, real size, real count)
{
	float2 ab = float2(a, b);
	float2 uv = opAB2UV(ab);
	real diag = size * real(sqrt_half) - uv.y;
	diag = r_fmod_const(diag, real(sqrt_two) * size / count);
	uv.y = size * real(sqrt_half) - diag;
	ab = opAB2UV(uv);

	real a_offset = (count - real(1.f)) / count;
	return length(float2(ab.x - a_offset * size, ab.y)) - size / count;
}
// :105-108
inline real opPipeMerge(real a, real b, real size, real count) { return r_min(r_min(a, b), opPipe(a, b, size, count)); }

// :112-115
inline real staircase(real x, real stepval, real spread)
{
	return r_min(stepval, r_frac(x / spread) * spread) + r_floor(x / spread) * stepval;
}

// :118-122
inline real smin(real a, real b, real k)
{
	real h = r_saturate(real(0.5f) + r_div_const(real(0.5f) * (b - a), k)); // census: numerator domain of the kernels' div_c
	return r_lerp(b, a, h) - k * h * (real(1.f) - h);
}
// :125-129
inline real smax1(real a, real b, real k)
{
	real h = r_saturate(real(0.5f) - real(0.5f) * (b + a) / k);
	return r_lerp(b, -a, h) + k * h * (real(1.f) - h);
}
// :131-135
inline real smax2(real a, real b, real k)
{
	real h = r_saturate(real(0.5f) - r_div_const(real(0.5f) * (b - a), k));
	return r_lerp(b, a, h) + k * h * (real(1.f) - h);
}

// ---- sdf_common.hlsl ---------------------------------------------------------------

// :4-10
inline float3 HUEtoRGB(real H)
{
	real R = r_abs(H * real(6.f) - real(3.f)) - real(1.f);
	real G = real(2.f) - r_abs(H * real(6.f) - real(2.f));
	real B = real(2.f) - r_abs(H * real(6.f) - real(4.f));
	return v_saturate(float3(R, G, B));
}
// :12-16
inline float3 HSVtoRGB(float3 HSV)
{
	float3 RGB = HUEtoRGB(HSV.x);
	return ((RGB - real(1.f)) * HSV.y + real(1.f)) * HSV.z;
}
// :18-22
inline real RGBtoBrightness(float3 rgb) { return dot(rgb, float3(real(0.2126f), real(0.7152f), real(0.0722f))); }

// :24-28
inline float2 get_tile_impact(float3 pos, float3 dir)
{
	real to_move = pos.y / dir.y;
	return float2(pos.x, pos.z) - float2(dir.x, dir.z) * to_move;
}
// :30-41
inline float4 tile_color_from_pos(float2 pos)
{
	float2 tile_index = v_floor(pos);
	float2 tile_pos = pos - tile_index;
	real tile_parity = r_round(r_frac((tile_index.x + tile_index.y) * real(0.5f) + real(0.25f)));
	float3 color = (tile_parity > real(0.5f)) ? float3(real(0.1f)) : float3(real(0.8f));

	float2 dist_vec = real(0.5f) - v_abs(tile_pos - real(0.5f));
	real dist = r_min(dist_vec.x, dist_vec.y);
	return float4(color, dist);
}
// :43-60
inline float3 total_tile_color(float3 pos, float3 dir, float3 offset_right, float3 offset_bottom)
{
	float4 color1 = tile_color_from_pos(get_tile_impact(pos, dir));
	float4 color2 = tile_color_from_pos(get_tile_impact(pos + offset_right, dir));
	float4 color3 = tile_color_from_pos(get_tile_impact(pos + offset_bottom, dir));
	float4 color4 = tile_color_from_pos(get_tile_impact(pos + offset_bottom + offset_right, dir));

	real total_dist = color1.w + color2.w + color3.w + color4.w;
	float3 color = (color1.xyz() * color1.w + color2.xyz() * color2.w + color3.xyz() * color3.w + color4.xyz() * color4.w) / total_dist;
	return color;
}
// :62-83
inline void map_groundplane(const GeometryInput &geometry, MaterialOutput &material_output, bool geometry_step, real &output_scene_distance)
{
	real floor1 = sdPlaneFast(geometry.pos, geometry.dir, float3(real(0.f), real(1.f), real(0.f)));
	if (geometry_step)
	{
		object_add(output_scene_distance, floor1);
	}
	else if (material_hit(floor1))
	{
		float3 offset_right = geometry.right_ray_offset * geometry.camera_distance;
		float3 offset_bottom = geometry.bottom_ray_offset * geometry.camera_distance;
		float3 color = total_tile_color(geometry.pos, geometry.dir.xyz(), offset_right, offset_bottom);
		material_output.diffuse_color = float4(color, real(1.f));
		material_output.specular_color.x = material_output.specular_color.y = material_output.specular_color.z = real(1.f);
	}
}
// :85-94
inline float3 sky_color(float3 dir, real phase)
{
	float2 rot = opRotate(float2(dir.x, dir.z), -phase * real(0.025f));
	dir.x = rot.x;
	dir.z = rot.y;
	real noiseval = turbulence(dir * float3(real(1.f), real(6.f), real(1.f)) * real(2.5f));
	float3 color1 = float3(real(43.f), real(164.f), real(247.f)) / real(255.f);
	float3 color2 = float3(real(212.f), real(224.f), real(238.f)) / real(255.f);
	float3 sky = lerp(color1, color2, noiseval) * real(1.2f);
	float3 horizon_color = float3(real(0.25f));
	return lerp(horizon_color, sky, r_saturate(dir.y * real(8.f) + real(0.125f)));
}

// ---- sdf_materials.hlsl ------------------------------------------------------------

// :6-13
inline float3 marble(float3 pos, float3 marble_color)
{
	float3 marble_dir = float3(real(3.f), real(2.f), real(1.f));
	real wave_pos = dot(marble_dir, pos) * real(2.f) + turbulence(pos) * real(5.f);
	real sine_val = (real(1.f) + r_sin(wave_pos)) * real(0.5f);
	sine_val = r_pow(sine_val, real(0.5f));
	return marble_color * sine_val;
}
// :15-24
inline float3 wood(float3 pos)
{
	const real turbulence_scale = 0.125f;
	const real rings = 12.f;
	real dist = r_sqrt(pos.x * pos.x + pos.y * pos.y) + turbulence_scale * turbulence(pos);
	real sine_val = real(0.5f) * r_abs(r_sin(real(2.f) * rings * dist * real(3.14159f)));
	return float3(real(0.3125f) + sine_val, real(0.117f) + sine_val, real(0.117f));
}
// :26-31
inline float4 fire(float3 pos, real threshold)
{
	real turb = turbulence(pos) + real(0.35f);
	turb = (turb > threshold) ? turb : real(0.f);
	return float4(real(5.f), real(2.f), real(1.f), real(0.5f)) * turb;
}
// :143-154
inline float3 debug_plane_color(real scene_distance)
{
	real int_steps;
	real frac_steps = r_abs(r_modf(scene_distance, int_steps)) * real(1.2f);
	real band_steps = r_modf(int_steps / real(5.f), int_steps);

	float3 band_color = (band_steps > real(0.7f)) ? float3(real(1.f), real(0.25f), real(0.25f)) : float3(real(0.75f), real(0.75f), real(1.f));
	frac_steps = (scene_distance < real(25.f)) ? frac_steps : real(0.5f);
	float3 col = (frac_steps < real(1.f)) ? frac_steps * frac_steps * float3(real(1.f)) : band_color;
	col.y = (scene_distance < real(0.f)) ? ((scene_distance > real(-0.01f)) ? real(1.f) : real(0.f)) : col.y;
	return col;
}
// :161-186
inline float3 iter_count_to_color(uint iter_count, uint max_iter_count)
{
	real rel_iter_count = real((float)iter_count) / real((float)max_iter_count);
	float3 col1 = float3(real(0.f), real(0.f), real(0.f));
	float3 col2 = float3(real(0.f), real(0.f), real(1.f));
	float3 col3 = float3(real(0.f), real(1.f), real(0.f));
	float3 col4 = float3(real(1.f), real(1.f), real(0.f));
	float3 col5 = float3(real(1.f), real(0.f), real(0.f));
	if (rel_iter_count < real(0.1f))
		return lerp(col1, col2, rel_iter_count / real(0.1f));
	else if (rel_iter_count < real(0.5f))
		return lerp(col2, col3, (rel_iter_count - real(0.1f)) / real(0.4f));
	else if (rel_iter_count < real(0.9f))
		return lerp(col3, col4, (rel_iter_count - real(0.5f)) / real(0.4f));
	else
		return lerp(col4, col5, (rel_iter_count - real(0.9f)) / real(0.1f));
}
// :194-201
inline real coordinate_material(float3 pos, float3 norm, real width)
{
	float3 reduced_pos = pos - v_floor(pos);
	reduced_pos = v_abs(reduced_pos - real(0.5f));
	float3 hits_tick = v_saturate((reduced_pos - real(0.5f) + width) * real(100.f));
	float3 mask = real(1.f) - v_abs(norm);
	return dot(hits_tick, mask);
}

} // namespace orc
