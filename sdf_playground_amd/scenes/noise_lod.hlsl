// noise_lod.hlsl -- a scene in the reference's dialect for the two corners of it no reference scene visits:
//  * simplex noise in 2 and 4 dimensions and grad4 (noise.hlsl:124-203, 304-433): an animated bump on a ball (4-D: position
//    and time), a banded slab (2-D), a lamp coloured by a gradient of the 4-D table;
//  * a geometry step that READS THE MARCH STATE the reference hands it (pshader_sdf.hlsl:187-218): the ball's bump is dropped
//    beyond `lod` units along the ray (level of detail by geometry.camera_distance), and a thin ring is thickened by the
//    pixel's footprint at that distance (geometry.right_ray_offset * geometry.camera_distance), so it never gets thinner
//    than a pixel.
// Oracle twin: SceneNoiseLod (oracle/test_scenes.h); tests/test_gpu_hlsl.py requires the same bits.
#include "sdf_primitives.hlsl"
#include "sdf_ops.hlsl"
#include "sdf_common.hlsl"
#include "noise.hlsl"

static const float3 ball_centre = float3(0.f, 1.2f, 0.f);
static const float4 grad_ip = float4(0.003401360544217687075f, 0.020408163265306122449f, 0.142857142857142857143f, 0.f);

float ball(float3 p, float camera_distance)
{
	float d = sdSphere(p - ball_centre, 1.f);
	if (camera_distance < VAR_lod(min = 0, max = 40, start = 9, step = 0.5))
	{
		float freq = VAR_freq(min = 0.5, max = 8, start = 3);
		d += VAR_bump(min = 0, max = 0.05, start = 0.02) * snoise(float4(p * freq, stime * 0.3f));
	}
	return d;
}

float slab(float3 p)
{
	return sdBox(p - float3(2.6f, 0.6f, 0.4f), float3(0.7f, 0.6f, 0.5f)) - 0.03f;
}

float ring(GeometryInput geometry)
{
	float footprint = length(geometry.right_ray_offset) * geometry.camera_distance;
	float3 q = geometry.pos - float3(-2.4f, 1.f, 0.3f);
	q.xz = opRotate(q.xz, 0.6f);
	return sdTorusXY(q, 0.8f, 0.01f + footprint);
}

float lamp(float3 p)
{
	return sdSphere(p - float3(0.8f, 2.9f, -1.2f), 0.15f);
}

void map(GeometryInput geometry, MarchingInput march, MaterialInput material_input, inout MaterialOutput material_output, bool geometry_step, inout float output_scene_distance)
{
	map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
	float d_ball = ball(geometry.pos, geometry.camera_distance);
	float d_slab = slab(geometry.pos);
	float d_ring = ring(geometry);
	float d_lamp = lamp(geometry.pos);
	if (geometry_step)
	{
		OBJECT(d_ball);
		OBJECT(d_slab);
		OBJECT(d_ring);
		if (!march.is_shadow_pass)
		{
			OBJECT(d_lamp);
		}
	}
	else
	{
		if (MATERIAL(d_ball))
		{
			float n = snoise(float4(geometry.pos * 1.5f, stime * 0.1f)) * 0.5f + 0.5f;
			material_output.diffuse_color = float4(lerp(float3(0.9f, 0.4f, 0.1f), float3(0.1f, 0.3f, 0.8f), n), 1.f);
			material_output.specular_color.rgb = 0.6f;
		}
		else if (MATERIAL(d_slab))
		{
			float bands = snoise(geometry.pos.xz * 4.f + float2(stime * 0.2f, 0.f));
			float fine = snoise(geometry.pos.xy * 17.f);
			material_output.diffuse_color.rgb = saturate(float3(0.5f, 0.5f, 0.5f) + bands * float3(0.4f, 0.1f, -0.3f) + fine * 0.08f);
			material_output.specular_color = float4(0.3f, 0.3f, 0.3f, 30.f);
			material_output.reflection_color = 0.15f;
		}
		else if (MATERIAL(d_ring))
		{
			material_output.diffuse_color = float4(0.9f, 0.8f, 0.2f, 1.f);
			material_output.specular_color.rgb = 1.f;
		}
		else if (MATERIAL(d_lamp))
		{
			float4 g = grad4(floor(stime * 3.f), grad_ip);
			material_output.emissive_color = abs(g.xyz) * 2.f + abs(g.w);
		}
	}
}

void map_normal(GeometryInput geometry, inout NormalOutput output)
{
}

void map_light(GeometryInput input, inout LightOutput output[LIGHT_COUNT], inout float ambient_lighting_factor)
{
	output[0].used = true;
	output[0].pos = float4(-1.f, -1.f, 2.f, 1.f);
	output[0].color = float3(1.f, 1.f, 1.f);
	output[1].used = true;
	output[1].pos.xyz = float3(0.8f, 2.9f, -1.2f);
	output[1].extend = 0.2f;
	output[1].falloff = 0.1f;
	output[1].color = float3(0.5f, 0.4f, 0.3f);
}

float3 map_background(float3 dir, uint iter_count)
{
	return sky_color(dir, stime);
}
