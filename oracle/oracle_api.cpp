// oracle/oracle_api.cpp -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// C entry points (ctypes) around the CPU restatement of the reference hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library; the product (sdf_playground_amd/) never does.
//
// PARITY STATUS: the reference hot path is HLSL executed by a D3D11 device and cannot
// be compiled or run here.  This oracle is pinned only by
//   * TestFastSphere1/5/6 (UnitTest/UnitTest.cpp:184-189,216-229) for sdSphereFast,
//   * TestStringSplit1..7 (UnitTest/UnitTest.cpp:91-179) for splitString,
//   * oracle/_ref (the reference's Math3D.cpp + Camera.cpp compiled here) for the
//     camera basis.
// Everything else on the path is "parity unpinned" by the reference: the oracle is a
// line-by-line restatement, self-checked with analytic known-answer tests.
#include "driver.h"
#include "host.h"
#include "postprocess.h"
#include "scenes.h"
#include "scenes2.h"
#include "scenes3.h"
#include "scenes4.h"
#include "test_scenes.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace orc;

extern "C" {

struct orc_frame
{
	float eye[3], front[3], right[3], top[3];
	float stime;
	int width, height;
	int iter_count, bounce_count, ray_count, light_count;
	float range;
	int max_cost_default;
	float debug_nx, debug_ny, debug_nz, debug_scale, debug_x, debug_y, debug_z, show_objects;
	float scene_var[8];
	int extension_lights; // 0..7, extension (SURVEY.md 8d cfg 5)
	float extension_marble_reflection; // 0 = reference, extension (SURVEY.md 8d cfg 3 as worded)
	float dist_eps, grad_eps, reflect_eps, refract_eps, shadow_eps; // pshader_sdf.hlsl:31-35
};

} // extern "C"

namespace {

struct SceneEntry
{
	const char *name;
	// VAR_ tags of the scene file in source order (data restated from the scene's HLSL;
	// e.g. scenes/sdf_scene_lense.hlsl:29-31,85)
	const char *var_decls;
	// scene variable names in the order Frame::scene_var uses
	std::vector<std::string> var_order;
	float4 (*pixel)(const Frame &, int, int, PixelStats &);
};

// VAR_ tags of the driver shader (pshader_sdf.hlsl:88-90,97-99,108,142)
const char *driver_var_decls =
	"VAR_debug_x(min = -10, max = +10, step = 0.02) VAR_debug_y(min = -10, max = +10, step = 0.02) "
	"VAR_debug_z(min = -10, max = +10, step = 0.02) VAR_debug_nx(min = -1, max = +1, step = 0.02) "
	"VAR_debug_ny(min = -1, max = +1, step = 0.02) VAR_debug_nz(min = -1, max = +1, step = 0.02) "
	"VAR_show_objects(min = 0, max = 1, step = 1, start = 1) VAR_debug_scale(min = 0.005, max = 2, step = 0.005, start = 0.2)";

const std::vector<SceneEntry> &scenes()
{
	static const std::vector<SceneEntry> table = {
		{"fast_sphere", "", {}, &ps_main<SceneFastSphere>},
		{"cube_sea", "", {}, &ps_main<SceneCubeSea>},
		{"labyrinth", "", {}, &ps_main<SceneLabyrinth>},
		{"fractal", "", {}, &ps_main<SceneFractal>},
		{"lense",
			"VAR_xpos(min = -4, max = 4, step = 0.1) VAR_ypos(min = -4, max = 4, step = 0.1) "
			"VAR_zpos(min = 0, max = 25, step = 0.1) VAR_mixing(min = 0, max = 1, step = 0.05)",
			{"xpos", "ypos", "zpos", "mixing"}, &ps_main<SceneLense>},
		{"gems", "", {}, &ps_main<SceneGems>},
		{"light_shadows", "", {}, &ps_main<SceneLightShadows>},
		{"cube",
			"VAR_size(min = 0.2, max = 2, start = 1, step = 0.2) VAR_xpos(min = -2, max = 2, start = 0, step = 0.1) "
			"VAR_ypos(min = -2, max = 2, start = 0, step = 0.1) VAR_zpos(min = -2, max = 2, start = 0, step = 0.1) "
			"VAR_red(min = 0, max = 1, start = 0.9, step = 0.05) VAR_green(min = 0, max = 1, start = 0.7, step = 0.05) "
			"VAR_blue(min = 0, max = 1, start = 0.2, step = 0.05)",
			{"size", "xpos", "ypos", "zpos", "red", "green", "blue"}, &ps_main<SceneCube>},
		{"gyroid", "", {}, &ps_main<SceneGyroid>},
		{"basic_transparency", "", {}, &ps_main<SceneBasicTransparency>},
		{"basic_clouds", "VAR_offset(min = -5, max = 5, step = 0.05)", {"offset"}, &ps_main<SceneBasicClouds>},
		{"coordinate_material",
			"VAR_boxoffset(min = 0, max = 2, step = 0.1, start = 2) VAR_spherical(min = 0, max = 1, step = 1, start = 0) "
			"VAR_thres(min=0,max=1,step=0.05, start=0.4)",
			{"boxoffset", "spherical", "thres"}, &ps_main<SceneCoordinateMaterial>},
		{"distortion", "", {}, &ps_main<SceneDistortion>},
		{"table", "", {}, &ps_main<SceneTable>},
		{"sierpinski", "", {}, &ps_main<SceneSierpinski>},
		{"neon",
			"VAR_r1(min = 0.2, max = 2, start = 1) VAR_r2(min = 0.005, max = 0.1, start = 0.01) VAR_spacing(min = 0.01, max = 0.2, start = 0.1) "
			"VAR_red(min = 0, max = 3, start = 0.1, step = 0.05) VAR_green(min = 0, max = 3, start = 1.0, step = 0.05) "
			"VAR_blue(min = 0, max = 3, start = 0.2, step = 0.05)",
			{"r1", "r2", "spacing", "red", "green", "blue"}, &ps_main<SceneNeon>},
		{"fractal2", "VAR_slider(min = -5, max = 5, step = 0.01, start = 0)", {"slider"}, &ps_main<SceneFractal2>},
		{"shell", "", {}, &ps_main<SceneShell>},
		{"spiral", "", {}, &ps_main<SceneSpiral>},
		{"terrain", "VAR_levels(min=1, max=10, step=1, start=2)", {"levels"}, &ps_main<SceneTerrain>},
		{"tiling",
			"VAR_m1(min = -1, max = 3, step = 0.1, start = 1) VAR_m2(min = -1, max = 3, step = 0.1, start = 0) "
			"VAR_width(min = 0.1, max = 0.5, step = 0.05, start = 0.4) VAR_run_length(min = 1, max = 10, step = 1, start = 4) "
			"VAR_run_flip(min = 1, max = 10, step = 1, start = 2) VAR_flip_chance(min = 0, max = 1, steps = 0.05) "
			"VAR_truchet_width(min = 0, max = 0.2, step = 0.01)",
			{"m1", "m2", "width", "run_length", "run_flip", "flip_chance", "truchet_width"}, &ps_main<SceneTiling>},
		{"tree", "", {}, &ps_main<SceneTree>},
	};
	return table;
}

// scenes that exist for tests only (test_scenes.h): found by name, not part of the reference's list
const std::vector<SceneEntry> &test_scenes()
{
	static const std::vector<SceneEntry> table = {
		{"debug_materials", "", {}, &ps_main<SceneDebugMaterials>},
		{"light_shadows_backwards", "", {}, &ps_main<SceneLightShadowsT<true>>}, // scenes.h: for Images/multi-lights.png only
		{"normal_test", "VAR_round(min = 0.0001, max = 0.05, start = 0.01) VAR_analytic(min = 0, max = 1, step = 1, start = 1)", {"round", "analytic"},
			&ps_main<SceneNormalTest>},
		// the VAR_ tags of sdf_playground_amd/scenes/noise_lod.hlsl and dialect_tour.hlsl, in the order of the text
		{"noise_lod", "VAR_lod(min = 0, max = 40, start = 9, step = 0.5) VAR_freq(min = 0.5, max = 8, start = 3) VAR_bump(min = 0, max = 0.05, start = 0.02)",
			{"lod", "freq", "bump"}, &ps_main<SceneNoiseLod>},
		{"dialect_tour", "VAR_spin(min = -2, max = 2, step = 0.1, start = 0.4) VAR_reach(min = 1, max = 3) VAR_blend(min = 0.01, max = 0.3, start = 0.08, steps = 7) "
			"VAR_shine() VAR_shine(min = 0, max = 1, start = 0.3, step = 0.05)", {"spin", "reach", "blend", "shine"}, &ps_main<SceneDialectTour>},
	};
	return table;
}

const SceneEntry *find_scene(const char *name)
{
	for (const auto &s : scenes())
		if (strcmp(s.name, name) == 0)
			return &s;
	for (const auto &s : test_scenes())
		if (strcmp(s.name, name) == 0)
			return &s;
	return nullptr;
}

Frame to_frame(const orc_frame &f)
{
	Frame F;
	F.eye = float3(f.eye[0], f.eye[1], f.eye[2]);
	F.front_vec = float3(f.front[0], f.front[1], f.front[2]);
	F.right_vec = float3(f.right[0], f.right[1], f.right[2]);
	F.top_vec = float3(f.top[0], f.top[1], f.top[2]);
	F.stime = f.stime;
	F.width = f.width;
	F.height = f.height;
	F.iter_count = f.iter_count;
	F.bounce_count = f.bounce_count;
	F.ray_count = f.ray_count;
	F.light_count = f.light_count;
	F.range = f.range;
	F.max_cost_default = (uint)f.max_cost_default;
	F.debug_nx = f.debug_nx;
	F.debug_ny = f.debug_ny;
	F.debug_nz = f.debug_nz;
	F.debug_scale = f.debug_scale;
	F.debug_x = f.debug_x;
	F.debug_y = f.debug_y;
	F.debug_z = f.debug_z;
	F.show_objects = f.show_objects;
	for (int i = 0; i < MAX_SCENE_VARS; ++i)
		F.scene_var[i] = f.scene_var[i];
	F.extension_lights = f.extension_lights < 0 ? 0 : (f.extension_lights > 7 ? 7 : f.extension_lights);
	F.extension_marble_reflection = f.extension_marble_reflection;
	F.dist_eps = f.dist_eps;
	F.grad_eps = f.grad_eps;
	F.reflect_eps = f.reflect_eps;
	F.refract_eps = f.refract_eps;
	F.shadow_eps = f.shadow_eps;
	return F;
}

int copy_out(const std::string &s, char *buf, int cap)
{
	if ((int)s.size() + 1 > cap)
		return -(int)s.size() - 1;
	memcpy(buf, s.c_str(), s.size() + 1);
	return (int)s.size();
}

} // namespace

extern "C" {

int orc_scene_count() { return (int)scenes().size(); }
const char *orc_scene_name(int i) { return (i >= 0 && i < (int)scenes().size()) ? scenes()[i].name : nullptr; }

// Variable table of a scene exactly as ShaderVariableManager would build it: driver
// tags + scene tags parsed, std::map order.  Output: one line per variable
// "name min max start step value scene_slot\n" (scene_slot = index into
// orc_frame.scene_var, or -1 for the driver's own variables).
int orc_var_table(const char *scene, char *buf, int cap)
{
	const SceneEntry *s = find_scene(scene);
	if (!s)
		return -1;
	host::VariableMap vars;
	std::string text = std::string(driver_var_decls) + " " + s->var_decls;
	if (!host::parseVariables(text, vars))
		return -2;
	std::string out;
	for (const auto &[name, var] : vars)
	{
		int slot = -1;
		for (size_t k = 0; k < s->var_order.size(); ++k)
			if (s->var_order[k] == name)
				slot = (int)k;
		char line[256];
		snprintf(line, sizeof line, "%s %.9g %.9g %.9g %.9g %.9g %d\n", name.c_str(), var.minval, var.maxval, var.start, var.step, var.value, slot);
		out += line;
	}
	return copy_out(out, buf, cap);
}

// Generic text front-end of the VAR_ parser (ShaderUtil.cpp:122-191)
int orc_parse_vars(const char *text, char *buf, int cap)
{
	host::VariableMap vars;
	if (!host::parseVariables(text, vars))
		return -2;
	std::string out;
	for (const auto &[name, var] : vars)
	{
		char line[256];
		snprintf(line, sizeof line, "%s %.9g %.9g %.9g %.9g %.9g\n", name.c_str(), var.minval, var.maxval, var.start, var.step, var.value);
		out += line;
	}
	return copy_out(out, buf, cap);
}

// splitString (Util.cpp:17-49): parts joined by \x1f, then \x1e, then separators joined by \x1f
int orc_split_string(const char *input, const char *pattern_start, const char *pattern_end, char *buf, int cap)
{
	auto [parts, seps] = host::splitString(input, pattern_start, pattern_end);
	std::string out;
	for (size_t i = 0; i < parts.size(); ++i)
	{
		if (i)
			out += '\x1f';
		out += std::string(parts[i]);
	}
	out += '\x1e';
	for (size_t i = 0; i < seps.size(); ++i)
	{
		if (i)
			out += '\x1f';
		out += std::string(seps[i]);
	}
	return copy_out(out, buf, cap);
}

int orc_remove_spaces(const char *input, char *buf, int cap) { return copy_out(std::string(host::removeSpaces(input)), buf, cap); }

// camera basis (eye, front, right, top) as 12 floats
void orc_camera_lookat(const float *eye, const float *lookat, float fovy, float aspect, float roll, float *out12)
{
	host::CameraBasis cb = host::camera_lookat(host::V3{eye[0], eye[1], eye[2]}, host::V3{lookat[0], lookat[1], lookat[2]}, fovy, aspect, roll);
	const host::V3 v[4] = {cb.eye, cb.front, cb.right, cb.top};
	for (int i = 0; i < 4; ++i)
	{
		out12[3 * i + 0] = v[i].x;
		out12[3 * i + 1] = v[i].y;
		out12[3 * i + 2] = v[i].z;
	}
}
void orc_camera_direction(const float *eye, const float *dir, float fovy, float aspect, float roll, float *out12)
{
	host::CameraBasis cb = host::camera_from_direction(host::V3{eye[0], eye[1], eye[2]}, host::V3{dir[0], dir[1], dir[2]}, fovy, aspect, roll);
	const host::V3 v[4] = {cb.eye, cb.front, cb.right, cb.top};
	for (int i = 0; i < 4; ++i)
	{
		out12[3 * i + 0] = v[i].x;
		out12[3 * i + 1] = v[i].y;
		out12[3 * i + 2] = v[i].z;
	}
}

// Render the pixels {(x, y): x = x0 + i*step_x < x1, y = y0 + j*step_y < y1} of a
// width x height frame.  out_rgba (4 floats/pixel) and out_stats (3 uint32/pixel:
// rays, march evals, hits; may be NULL) are indexed by the full-frame pixel index
// y*width + x; untouched pixels keep their previous content.  totals (may be NULL)
// receives {pixels, rays, march_evals, hits} as 4 uint64.  Rows are distributed over
// nthreads threads (dynamic, one row at a time).
int orc_render(const char *scene, const orc_frame *frame, float *out_rgba, unsigned *out_stats, int x0, int y0, int x1, int y1,
	int step_x, int step_y, int nthreads, unsigned long long *totals)
{
	const SceneEntry *s = find_scene(scene);
	if (!s)
		return -1;
	if (frame->ray_count < 1 || frame->ray_count > MAX_RAY_COUNT || frame->light_count < 0 || frame->light_count > MAX_LIGHT_COUNT ||
		frame->width < 1 || frame->height < 1 || step_x < 1 || step_y < 1 || frame->iter_count < 1 || frame->bounce_count < 0)
		return -2;
	if (!(frame->dist_eps > 0.f) || !(frame->grad_eps > 0.f) || !(frame->reflect_eps >= 0.f) || !(frame->refract_eps >= 0.f) || !(frame->shadow_eps >= 0.f))
		return -3; // a frame that was not made by default_frame
	const Frame F = to_frame(*frame);
	// the driver's epsilons (sdf_lib.h): set before the threads start
	dist_eps = F.dist_eps;
	grad_eps = F.grad_eps;
	reflect_eps = F.reflect_eps;
	refract_eps = F.refract_eps;
	shadow_eps = F.shadow_eps;
	if (nthreads < 1)
		nthreads = 1;
	std::atomic<int> next_row(0);
	const int nrows = (y1 - y0 + step_y - 1) / step_y;
	std::vector<unsigned long long> tot((size_t)nthreads * 4, 0ull);
	auto worker = [&](int tid) {
		unsigned long long t[4] = {0, 0, 0, 0};
		for (;;)
		{
			int r = next_row.fetch_add(1);
			if (r >= nrows)
				break;
			int y = y0 + r * step_y;
			for (int x = x0; x < x1; x += step_x)
			{
				PixelStats st = {0, 0, 0};
				float4 c = s->pixel(F, x, y, st);
				size_t idx = (size_t)y * (size_t)F.width + (size_t)x;
				out_rgba[4 * idx + 0] = val(c.x);
				out_rgba[4 * idx + 1] = val(c.y);
				out_rgba[4 * idx + 2] = val(c.z);
				out_rgba[4 * idx + 3] = val(c.w);
				if (out_stats)
				{
					out_stats[3 * idx + 0] = st.rays;
					out_stats[3 * idx + 1] = st.march_evals;
					out_stats[3 * idx + 2] = st.hits;
				}
				t[0] += 1;
				t[1] += st.rays;
				t[2] += st.march_evals;
				t[3] += st.hits;
			}
		}
		for (int k = 0; k < 4; ++k)
			tot[(size_t)tid * 4 + k] = t[k];
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < nthreads; ++t)
		pool.emplace_back(worker, t);
	worker(0);
	for (auto &th : pool)
		th.join();
	if (totals)
	{
		for (int k = 0; k < 4; ++k)
		{
			totals[k] = 0;
			for (int t = 0; t < nthreads; ++t)
				totals[k] += tot[(size_t)t * 4 + k];
		}
	}
	return 0;
}

#ifdef ORACLE_CENSUS
// flop census of the calling thread (render with nthreads = 1): {flops, transcendentals}
void orc_census_reset() { census() = Census{0, 0, 0, 0, 0}; }
// {flops, transcendentals, near-field sqrt / division arguments out of domain, far-field (overflowed) arguments}
void orc_census_get(unsigned long long *out4)
{
	out4[0] = census().flops;
	out4[1] = census().transc;
	out4[2] = census().sqrt_out_of_domain;
	out4[3] = census().divc_out_of_domain;
	out4[4] = census().far_field;
}
#endif

// HDR::process restated (oracle/postprocess.h).  scene: RGBA16F [h][w][4]; bloom1/bloom2:
// RGBA16F scratch/outputs (may be inspected by tests); ldr: RGBA8 [h][w][4].
void orc_postprocess(const unsigned short *scene, int w, int h, unsigned short *bloom1, unsigned short *bloom2, unsigned char *ldr)
{
	post::Image16 s = {w, h, scene};
	post::bloom_horizontal(s, bloom1);
	post::Image16 b1 = {w, h, bloom1};
	post::bloom_vertical(b1, bloom2);
	post::Image16 b2 = {w, h, bloom2};
	post::tonemap(s, b2, ldr);
}
void orc_float_to_half(const float *in, unsigned short *out, long long n)
{
	for (long long i = 0; i < n; ++i) out[i] = post::float_to_half(in[i]);
}
void orc_half_to_float(const unsigned short *in, float *out, long long n)
{
	for (long long i = 0; i < n; ++i) out[i] = post::half_to_float(in[i]);
}

// the oracle's simplex noise, many points per call: what = 2 / 3 / 4 dimensions (in: n x what floats, out: n floats),
// 5 = grad4 (in: n x 4 floats j, ip.xyz; out: n x 4)
void orc_noise(int what, const float *in, float *out, long long n)
{
	for (long long k = 0; k < n; ++k)
	{
		if (what == 2) out[k] = val(snoise(float2(in[2 * k], in[2 * k + 1])));
		else if (what == 3) out[k] = val(snoise(float3(in[3 * k], in[3 * k + 1], in[3 * k + 2])));
		else if (what == 4) out[k] = val(snoise(float4(in[4 * k], in[4 * k + 1], in[4 * k + 2], in[4 * k + 3])));
		else if (what == 5)
		{
			float4 g = grad4(in[4 * k], float4(in[4 * k + 1], in[4 * k + 2], in[4 * k + 3], 0.0f));
			out[4 * k] = val(g.x); out[4 * k + 1] = val(g.y); out[4 * k + 2] = val(g.z); out[4 * k + 3] = val(g.w);
		}
	}
}

// Known-answer access to individual library functions (tests/test_oracle_*.py).
// Returns the number of outputs written, or -1 for an unknown function.
int orc_kat(const char *fn, const float *in, float *out)
{
	std::string f(fn);
	auto v3 = [&](int o) { return float3(in[o], in[o + 1], in[o + 2]); };
	auto put3 = [&](float3 v) { out[0] = val(v.x); out[1] = val(v.y); out[2] = val(v.z); return 3; };
	if (f == "sdSphereFast") { out[0] = val(sdSphereFast(v3(0), float4(v3(3), in[6]), in[7])); return 1; }
	if (f == "sdSphere") { out[0] = val(sdSphere(v3(0), in[3])); return 1; }
	if (f == "sdBox") { out[0] = val(sdBox(v3(0), v3(3))); return 1; }
	if (f == "sdPlane") { out[0] = val(sdPlane(v3(0), v3(3))); return 1; }
	if (f == "sdPlaneFast") { out[0] = val(sdPlaneFast(v3(0), float4(v3(3), in[6]), v3(7))); return 1; }
	if (f == "sdTorusXY") { out[0] = val(sdTorusXY(v3(0), in[3], in[4])); return 1; }
	if (f == "sdCappedCylinder") { out[0] = val(sdCappedCylinder(v3(0), in[3], in[4])); return 1; }
	if (f == "sdRoundCone") { out[0] = val(sdRoundCone(v3(0), v3(3), v3(6), in[9], in[10])); return 1; }
	if (f == "sdLimit2") { out[0] = val(sdLimit2(float2(in[0], in[1]), float2(in[2], in[3]), float2(in[4], in[5]))); return 1; }
	if (f == "opRepInf") { out[0] = val(opRepInf(real(in[0]), real(in[1]))); return 1; }
	if (f == "opRepLim") { out[0] = val(opRepLim(real(in[0]), real(in[1]), real(in[2]))); return 1; }
	if (f == "opRepAngle") { float2 p(in[0], in[1]); out[2] = val(opRepAngle(p, in[2])); out[0] = val(p.x); out[1] = val(p.y); return 3; }
	if (f == "opRotate") { float2 p = opRotate(float2(in[0], in[1]), in[2]); out[0] = val(p.x); out[1] = val(p.y); return 2; }
	if (f == "opPipe") { out[0] = val(opPipe(in[0], in[1], in[2], in[3])); return 1; }
	if (f == "smin") { out[0] = val(smin(in[0], in[1], in[2])); return 1; }
	if (f == "smax2") { out[0] = val(smax2(in[0], in[1], in[2])); return 1; }
	if (f == "round") { out[0] = val(r_round(in[0])); return 1; }
	if (f == "fmod") { out[0] = val(r_fmod(in[0], in[1])); return 1; }
	if (f == "frac") { out[0] = val(r_frac(in[0])); return 1; }
	if (f == "sign") { out[0] = val(r_sign(in[0])); return 1; }
	if (f == "step") { out[0] = val(r_step(in[0], in[1])); return 1; }
	if (f == "min") { out[0] = val(r_min(in[0], in[1])); return 1; }
	if (f == "max") { out[0] = val(r_max(in[0], in[1])); return 1; }
	if (f == "pow") { out[0] = val(r_pow(in[0], in[1])); return 1; }
	if (f == "sin") { out[0] = val(r_sin(in[0])); return 1; }
	if (f == "cos") { out[0] = val(r_cos(in[0])); return 1; }
	if (f == "atan2") { out[0] = val(r_atan2(in[0], in[1])); return 1; }
	if (f == "exp2") { out[0] = val(r_exp2(in[0])); return 1; }
	if (f == "log2") { out[0] = val(r_log2(in[0])); return 1; }
	if (f == "modf") { real ip; out[0] = val(r_modf(in[0], ip)); out[1] = val(ip); return 2; }
	if (f == "reflect") return put3(reflect(v3(0), v3(3)));
	if (f == "refract") return put3(refract(v3(0), v3(3), in[6]));
	if (f == "normalize") return put3(normalize(v3(0)));
	if (f == "hash") { uint32_t u; memcpy(&u, &in[0], 4); uint32_t h = hash(u); memcpy(&out[0], &h, 4); return 1; }
	if (f == "hashf") { uint32_t u; memcpy(&u, &in[0], 4); out[0] = val(hashf(u)); return 1; }
	if (f == "snoise3") { out[0] = val(snoise(v3(0))); return 1; }
	if (f == "snoise2") { out[0] = val(snoise(float2(in[0], in[1]))); return 1; }
	if (f == "snoise4") { out[0] = val(snoise(float4(in[0], in[1], in[2], in[3]))); return 1; }
	if (f == "grad4") { float4 g = grad4(in[0], float4(in[1], in[2], in[3], in[4])); out[0] = val(g.x); out[1] = val(g.y); out[2] = val(g.z); out[3] = val(g.w); return 4; }
	if (f == "permute") { out[0] = val(permute(real(in[0]))); return 1; }
	if (f == "mod289") { out[0] = val(mod289(real(in[0]))); return 1; }
	if (f == "turbulence") { out[0] = val(turbulence(v3(0))); return 1; }
	if (f == "tile_color") { float4 c = tile_color_from_pos(float2(in[0], in[1])); out[0] = val(c.x); out[1] = val(c.y); out[2] = val(c.z); out[3] = val(c.w); return 4; }
	if (f == "sky_color") return put3(sky_color(v3(0), in[3]));
	if (f == "marble") return put3(marble(v3(0), v3(3)));
	if (f == "wood") return put3(wood(v3(0)));
	if (f == "fire") { float4 c = fire(v3(0), in[3]); out[0] = val(c.x); out[1] = val(c.y); out[2] = val(c.z); out[3] = val(c.w); return 4; }
	if (f == "debug_plane_color") return put3(debug_plane_color(in[0]));
	if (f == "iter_count_to_color") return put3(iter_count_to_color((uint)in[0], (uint)in[1]));
	if (f == "HSVtoRGB") return put3(HSVtoRGB(v3(0)));
	return -1;
}

} // extern "C"
