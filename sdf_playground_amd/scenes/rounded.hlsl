// rounded.hlsl -- a scene in the reference's dialect that uses the callback every scene of the reference leaves empty:
// map_normal (sdf_structs.hlsl:39-52).  A ball with its analytic normal (use_normal), a mirror-coated block and a drum whose
// normals are sampled wider apart (the variable "round") ("larger than usual values lead to rounded corners"), a block left alone.  The same
// objects, materials and rules as the library's diagnostic scene "normal_test" (csrc/sdfr_scene_debug.h, oracle twin
// oracle/test_scenes.h): the three must render the same bits.
#include "sdf_primitives.hlsl"
#include "sdf_common.hlsl"

static const float3 ball_centre = float3(-1.6f, 0.7f, 0.2f);

float ball(float3 p) { return sdSphere(p - ball_centre, 0.7f); }
float block(float3 p) { return sdBox(p - float3(0.f, 0.5f, 0.f), float3(0.5f, 0.5f, 0.5f)); }
float drum(float3 p) { return sdCappedCylinder(p - float3(1.5f, 0.45f, -0.3f), 0.45f, 0.4f); }
float plain_block(float3 p) { return sdBox(p - float3(0.4f, 0.3f, -1.6f), float3(0.3f, 0.3f, 0.3f)); }

void map(GeometryInput geometry, MarchingInput march, MaterialInput material_input, inout MaterialOutput material_output, bool geometry_step, inout float output_scene_distance)
{
	map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
	float d_ball = ball(geometry.pos);
	float d_block = block(geometry.pos);
	float d_drum = drum(geometry.pos);
	float d_plain = plain_block(geometry.pos);
	if (geometry_step)
	{
		OBJECT(d_ball);
		OBJECT(d_block);
		OBJECT(d_drum);
		OBJECT(d_plain);
	}
	else if (MATERIAL(d_ball))
	{
		material_output.diffuse_color = float4(0.8f, 0.3f, 0.2f, 1.f);
		material_output.specular_color = float4(1.f, 1.f, 1.f, 20.f);
	}
	else if (MATERIAL(d_block))
	{
		material_output.diffuse_color = float4(0.2f, 0.3f, 0.8f, 1.f);
		material_output.specular_color.rgb = 0.5f;
		material_output.reflection_color = 0.4f;
	}
	else if (MATERIAL(d_drum))
	{
		material_output.material_id = MATERIAL_NORMAL2;
	}
	else if (MATERIAL(d_plain))
	{
		material_output.diffuse_color = float4(0.3f, 0.8f, 0.3f, 1.f);
		material_output.specular_color.rgb = 0.5f;
	}
}

void map_normal(GeometryInput geometry, inout NormalOutput output)
{
	float round_by = VAR_round(min = 0.0001, max = 0.05, start = 0.01);
	float analytic = VAR_analytic(min = 0, max = 1, step = 1, start = 1);
	if (analytic != 0.f && abs(ball(geometry.pos)) < 0.01f)
	{
		output.use_normal = true;
		output.normal = normalize(geometry.pos - ball_centre);
	}
	else if (abs(block(geometry.pos)) < 0.01f || abs(drum(geometry.pos)) < 0.01f)
	{
		output.normal_sample_dist = round_by;
	}
}

void map_light(GeometryInput input, inout LightOutput output[LIGHT_COUNT], inout float ambient_lighting_factor)
{
	output[0].used = true;
	output[0].pos = float4(-1.f, -1.f, 2.f, 1.f);
	output[0].color = float3(1.f, 1.f, 1.f);
}

float3 map_background(float3 dir, uint iter_count)
{
	return sky_color(dir, stime);
}
