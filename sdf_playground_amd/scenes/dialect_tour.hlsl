// dialect_tour.hlsl -- one scene through the parts of the reference's scene dialect that its 22 scenes use between them
// (SURVEY.md 8 a-T.6-8), each where a pixel depends on it:
//   swizzled l-values and swaps (`p.yxz = sorted(p.yxz)`, `q.xz = q.zx`, `c.rgb = 0.5f`), an inout parameter handed a swizzle
//   (`opRepAngle(p.xz, 6.f)`: copy in, copy out), float3x3 + mul, `static const` globals initialised by cos / sin / normalize,
//   `(int)` with D3D's saturation feeding hashf, a scene-local overload of the library's voronoi, opRepLim / opShell / smin,
//   and variable tags with every key, with none, with an unknown key and with a name declared twice (the last one counts:
//   ShaderUtil.cpp:122-191).  (The tag prefix is not spelled out in these comments: the reference's parser reads comments too.)
// Oracle twin: SceneDialectTour (oracle/test_scenes.h); tests/test_gpu_hlsl.py requires the same bits.
#include "sdf_primitives.hlsl"
#include "sdf_ops.hlsl"
#include "sdf_common.hlsl"
#include "sdf_materials.hlsl"

static const float tilt = 25.f * pi / 180.f;
static const float tilt_c = cos(tilt), tilt_s = sin(tilt);
static const float3 slope_normal = normalize(float3(0.3f, 1.f, -0.2f));

float3 sorted(float3 v)
{
	// ascending by three compare-and-swaps, each a swizzled swap
	if (v.x > v.y) v.xy = v.yx;
	if (v.y > v.z) v.yz = v.zy;
	if (v.x > v.y) v.xy = v.yx;
	return v;
}

// the library's voronoi takes (uv, max_offset): one argument less is another function in HLSL, and the library's stays visible
float4 voronoi(float2 uv)
{
	return voronoi(uv * 3.f, 0.4f);
}

float carousel(float3 p, out float index)
{
	p -= float3(0.f, 0.9f, 0.f);
	p.xz = opRotate(p.xz, stime * VAR_spin(min = -2, max = 2, step = 0.1, start = 0.4));
	index = opRepAngle(p.xz, 6.f);
	p.x -= VAR_reach(min = 1, max = 3);
	float3x3 lean = { 1.f, 0.f, 0.f,
	                  0.f, tilt_c, -tilt_s,
	                  0.f, tilt_s, tilt_c };
	p = mul(lean, p);
	// a box whose half-sizes are the sorted |p|-independent constants: exercise the sort on a literal
	float3 half_size = sorted(float3(0.35f, 0.15f, 0.25f));
	return sdBox(p, half_size.zxy) - 0.04f;
}

float bowl(float3 p)
{
	p -= float3(0.f, 0.55f, 0.f);
	float shell = opShell(sdSphere(p, 0.5f), 0.05f, 0.f);
	float cut = sdPlane(p - float3(0.f, 0.1f, 0.f), slope_normal);
	return max(shell, cut);
}

float studs(float3 p)
{
	float3 q = p - float3(0.f, 0.05f, 3.f);
	q.xz = opRepLim(q.xz, float2(3.f, 1.f), float2(0.8f, 0.8f));
	float stud = sdCappedCylinder(q, 0.05f, 0.15f);
	float cap = sdSphere(q - float3(0.f, 0.12f, 0.f), 0.12f);
	return smin(stud, cap, VAR_blend(min = 0.01, max = 0.3, start = 0.08, steps = 7));
}

void map(GeometryInput geometry, MarchingInput march, MaterialInput material_input, inout MaterialOutput material_output, bool geometry_step, inout float output_scene_distance)
{
	map_groundplane(geometry, material_output, geometry_step, output_scene_distance);
	float index;
	float d_carousel = carousel(geometry.pos, index);
	float d_bowl = bowl(geometry.pos);
	float d_studs = studs(geometry.pos);
	if (geometry_step)
	{
		OBJECT(d_carousel);
		OBJECT(d_bowl);
		OBJECT(d_studs);
	}
	else
	{
		if (MATERIAL(d_carousel))
		{
			// a colour per arm: hashf of an int made from a float far outside the int range for odd arms (D3D saturates)
			float big = frac(index * 0.5f + 0.25f) > 0.5f ? 1e12f : 3.7f;
			float h = hashf((int)(index * 17.f + big));
			material_output.diffuse_color = float4(HSVtoRGB(float3(h, 0.8f, 0.9f)), 1.f);
			material_output.specular_color.rgb = 0.5f;
			material_output.reflection_color = VAR_shine();
		}
		else if (MATERIAL(d_bowl))
		{
			float4 cell = voronoi(geometry.pos.xz + geometry.pos.yy);
			float3 c;
			c.rgb = 0.5f;
			c.rg += float2(hashf((int)cell.x), hashf((int)cell.y)) * 0.5f;
			c.b *= smoothstep(0.02f, 0.1f, cell.w) + 0.2f;
			material_output.diffuse_color.xyz = c.bgr;
			material_output.specular_color = float4(1.f, 1.f, 1.f, 40.f);
		}
		else if (MATERIAL(d_studs))
		{
			material_output.diffuse_color = float4(0.7f, 0.7f, 0.75f, 1.f);
			material_output.specular_color.rgb = 1.f;
			// the same name again: this declaration is the one the variable table keeps
			material_output.reflection_color = VAR_shine(min = 0, max = 1, start = 0.3, step = 0.05);
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
	ambient_lighting_factor = 0.1f;
}

float3 map_background(float3 dir, uint iter_count)
{
	return sky_color(dir, stime);
}
