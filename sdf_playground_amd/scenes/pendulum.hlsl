// pendulum.hlsl -- the example scene of this directory in the REFERENCE'S OWN DIALECT (the scene plugin interface of
// Gotbread/sdf-playground: map / map_normal / map_light / map_background, OBJECT / MATERIAL macros, HLSL vector types):
// checker floor, a mirror ball swinging on a rod from a wooden gallows, one sun.  Twin of pendulum.scene.h: both render the
// same bits (tests/test_gpu_hlsl.py).  Load with sdfr_load_scene_hlsl / SDFRenderer.initShaderHlsl, or
//   python -m sdf_playground_amd.cli --scene-hlsl sdf_playground_amd/scenes/pendulum.hlsl --out pendulum.png
#include "sdf_primitives.hlsl"
#include "sdf_ops.hlsl"
#include "sdf_common.hlsl"

static const float3 pivot = float3(0.f, 3.f, 0.f);

// position in the pendulum's frame: pivot at the origin, rod along -y
float3 swing_frame(float3 p)
{
	float angle = VAR_swing(min = 0, max = 1.2, start = 0.7) * sin(stime * 1.5f);
	float3 q = p - pivot;
	q.xy = opRotate(q.xy, angle);
	return q;
}

float ball(float3 p)
{
	float len = VAR_rod(min = 0.5, max = 2.5, start = 1.8);
	return sdSphere(swing_frame(p) + float3(0.f, len, 0.f), VAR_radius(min = 0.1, max = 0.8, start = 0.45));
}

float rod(float3 p)
{
	float len = VAR_rod(min = 0.5, max = 2.5, start = 1.8);
	return sdCappedCylinder(swing_frame(p) + float3(0.f, len * 0.5f, 0.f), len * 0.5f, 0.03f);
}

float gallows(float3 p)
{
	float post = sdBox(p - float3(-1.5f, 1.6f, 0.f), float3(0.1f, 1.6f, 0.1f));
	float beam = sdBox(p - float3(-0.6f, 3.1f, 0.f), float3(1.f, 0.1f, 0.1f));
	return opChamferMerge(post, beam, 0.1f);
}

void map(GeometryInput geometry, MarchingInput march, MaterialInput material_input, inout MaterialOutput material_output, bool geometry_step, inout float output_scene_distance)
{
	map_groundplane(geometry, material_output, geometry_step, output_scene_distance);

	float d_ball = ball(geometry.pos);
	float d_rod = rod(geometry.pos);
	float d_gallows = gallows(geometry.pos);

	if (geometry_step)
	{
		OBJECT(d_ball);
		OBJECT(d_rod);
		OBJECT(d_gallows);
	}
	else
	{
		if (MATERIAL(d_ball))
		{
			material_output.diffuse_color = float4(0.05f, 0.05f, 0.08f, 1.f);
			material_output.specular_color.rgb = 1.f;
			material_output.reflection_color = 0.7f;
		}
		if (MATERIAL(d_rod))
		{
			material_output.diffuse_color = float4(0.6f, 0.6f, 0.65f, 1.f);
			material_output.specular_color.rgb = 1.f;
		}
		if (MATERIAL(d_gallows))
		{
			material_output.material_id = MATERIAL_WOOD;
			material_output.material_position.xyz = geometry.pos * 2.f;
			material_output.diffuse_color = float4(0.f, 0.f, 0.f, 1.f);
			material_output.specular_color.rgb = 0.2f;
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
}

float3 map_background(float3 dir, uint iter_count)
{
	return sky_color(dir, stime);
}
