// sdfr_api.cpp -- implementation of the C ABI declared in include/sdfr.h.
//
// Host logic only: handle state (scene, variable table, camera, limits), per-frame uniform
// preparation, device buffers and launches.  All pixels are produced by the HIP kernels in
// sdfr_kernels.hip; there is no CPU rendering path.
#include "sdfr_handle.h"
#include "sdfr_hlsl_translate.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

using namespace sdfr;

// VAR_ tags of the raymarch driver itself (reference: pshader_sdf.hlsl:88-90,97-99,108,142)
static const char *k_driver_variables =
	"VAR_debug_x(min = -10, max = +10, step = 0.02) VAR_debug_y(min = -10, max = +10, step = 0.02) "
	"VAR_debug_z(min = -10, max = +10, step = 0.02) VAR_debug_nx(min = -1, max = +1, step = 0.02) "
	"VAR_debug_ny(min = -1, max = +1, step = 0.02) VAR_debug_nz(min = -1, max = +1, step = 0.02) "
	"VAR_show_objects(min = 0, max = 1, step = 1, start = 1) VAR_debug_scale(min = 0.005, max = 2, step = 0.005, start = 0.2)";


namespace sdfr {
size_t image_bytes(size_t pixels, int format)
{
	if (format == SDFR_RGBA32F) return pixels * 16;
	if (format == SDFR_RGBA16F) return pixels * 8;
	if (format == SDFR_STRIP_RGB16F_A8) return (pixels * 7 + 3) & ~(size_t)3;
	return (pixels * 13 + 3) & ~(size_t)3;
}
bool is_wire_format(int format)
{
	return format == SDFR_RGBA32F || format == SDFR_RGBA16F || format == SDFR_STRIP_RGB32F_A8 || format == SDFR_STRIP_RGB16F_A8;
}
} // namespace sdfr

static void free_workspace(sdfr_renderer *r)
{
	WavefrontWorkspace &w = r->ws;
	(void)hipFree(w.ray_cur);
	(void)hipFree(w.ray_queue);
	(void)hipFree(w.qdepth_lo);
	(void)hipFree(w.qdepth_hi);
	(void)hipFree(w.result);
	(void)hipFree(w.accum);
	(void)hipFree(w.list_a);
	(void)hipFree(w.list_b);
	(void)hipFree(w.counters);
	(void)hipFree(w.partials);
	(void)hipFree(w.tile_cursors);
	w = WavefrontWorkspace{};
	r->wavefront_capacity = 0;
}

// Per-pixel scratch sized for `pixels` work items.  Both schedules use the pending-ray queue and
// the counter partials; the per-round state of the wavefront schedule (another 116 B per pixel)
// is allocated only once that schedule is used.  On failure everything is released (hipFree
// waits for the device, so buffers of frames still in flight are safe to drop).
static int ensure_workspace(sdfr_renderer *r, size_t pixels, bool wavefront)
{
	WavefrontWorkspace &w = r->ws;
	// the pixel kernel indexes the pending-ray records with 32 bits (GlobalRayStore::record)
	if (pixels * (size_t)SDFR_MAX_RAYS >= ((size_t)1 << 32)) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "frame too large: more than 2^32 / 8 pixels per launch");
	if (w.capacity < pixels || (wavefront && r->wavefront_capacity < pixels))
	{
		if (w.capacity < pixels) free_workspace(r);
		const size_t n = w.capacity < pixels ? pixels : w.capacity;
		hipError_t e = hipSuccess;
		auto alloc = [&](void **p, size_t bytes) {
			if (e == hipSuccess && *p == nullptr) e = hipMalloc(p, bytes);
		};
		alloc((void **)&w.ray_queue, sizeof(float) * 12 * SDFR_MAX_RAYS * n); // 48-byte records (pixel schedule) / 11 field arrays (wavefront)
		alloc((void **)&w.partials, sizeof(RenderTotals) * (n / 64 + 1));
		if (e == hipSuccess && w.tile_cursors == nullptr)
		{
			alloc((void **)&w.tile_cursors, sizeof(uint32_t) * (size_t)pixel_tile_cursor_words());
			if (e == hipSuccess) e = hipMemset(w.tile_cursors, 0, sizeof(uint32_t) * (size_t)pixel_tile_cursor_words());
		}
		if (wavefront)
		{
			alloc((void **)&w.ray_cur, sizeof(float) * 11 * n);
			alloc((void **)&w.qdepth_lo, sizeof(uint32_t) * n);
			alloc((void **)&w.qdepth_hi, sizeof(uint32_t) * n);
			alloc((void **)&w.result, sizeof(float) * 8 * n);
			alloc((void **)&w.accum, sizeof(float) * 4 * n);
			alloc((void **)&w.list_a, sizeof(uint32_t) * n);
			alloc((void **)&w.list_b, sizeof(uint32_t) * n);
			alloc((void **)&w.counters, sizeof(uint32_t) * 64);
		}
		if (e != hipSuccess)
		{
			free_workspace(r);
			return hip_fail(r, e, "workspace allocation");
		}
		w.capacity = n;
		if (wavefront) r->wavefront_capacity = n;
		w.pstat = nullptr;
	}
	return SDFR_OK;
}

// ---- two frames in flight (sdfr_set_frames_in_flight): see sdfr_renderer::Lane -----------------------------------
static void swap_lanes(sdfr_renderer *r)
{
	std::swap(r->stream, r->other.stream);
	std::swap(r->ws, r->other.ws);
	std::swap(r->wavefront_capacity, r->other.wavefront_capacity);
	std::swap(r->d_totals, r->other.d_totals);
	std::swap(r->totals_parts, r->other.totals_parts);
	std::swap(r->ev_begin, r->other.ev_begin);
	std::swap(r->ev_end, r->other.ev_end);
	std::swap(r->have_render, r->other.have_render);
	std::swap(r->out_lo, r->other.out_lo);
	std::swap(r->out_hi, r->other.out_hi);
}
static void release_second_lane(sdfr_renderer *r)
{
	if (r->frames_in_flight != 2) return;
	(void)hipStreamSynchronize(r->stream);
	(void)hipStreamSynchronize(r->other.stream);
	swap_lanes(r); // free_workspace works on the current lane
	free_workspace(r);
	swap_lanes(r);
	(void)hipFree(r->other.d_totals);
	if (r->other.ev_begin) (void)hipEventDestroy(r->other.ev_begin);
	if (r->other.ev_end) (void)hipEventDestroy(r->other.ev_end);
	r->other = sdfr_renderer::Lane();
	for (hipStream_t &s : r->lane_streams)
	{
		if (s) (void)hipStreamDestroy(s);
		s = nullptr;
	}
	r->stream = r->user_stream;
	r->frames_in_flight = 1;
}

extern "C" {

int sdfr_set_frames_in_flight(sdfr_renderer *r, int n)
{
	return guarded(r, [&]() -> int {
		if (!r || (n != 1 && n != 2)) return SDFR_ERR_INVALID_ARGUMENT;
		SDFR_HIP(hipSetDevice(r->device));
		if (n == r->frames_in_flight) return SDFR_OK;
		if (n == 1)
		{
			release_second_lane(r);
			return SDFR_OK;
		}
		SDFR_HIP(hipStreamSynchronize(r->stream));
		sdfr_renderer::Lane lane;
		hipError_t e = hipStreamCreateWithFlags(&r->lane_streams[0], hipStreamNonBlocking);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->lane_streams[1], hipStreamNonBlocking);
		if (e == hipSuccess) e = hipMalloc((void **)&lane.d_totals, 2 * sizeof(RenderTotals));
		if (e == hipSuccess) e = hipEventCreate(&lane.ev_begin);
		if (e == hipSuccess) e = hipEventCreate(&lane.ev_end);
		if (e != hipSuccess)
		{
			(void)hipFree(lane.d_totals);
			if (lane.ev_begin) (void)hipEventDestroy(lane.ev_begin);
			if (lane.ev_end) (void)hipEventDestroy(lane.ev_end);
			for (hipStream_t &s : r->lane_streams)
			{
				if (s) (void)hipStreamDestroy(s);
				s = nullptr;
			}
			return hip_fail(r, e, "sdfr_set_frames_in_flight");
		}
		lane.stream = r->lane_streams[1];
		r->other = lane;
		r->stream = r->lane_streams[0]; // the current lane keeps its workspace and the row order it has learned
		r->out_lo = r->out_hi = nullptr;
		r->frames_in_flight = 2;
		return SDFR_OK;
	});
}

int sdfr_wait_frame(sdfr_renderer *r, void *hip_stream)
{
	return guarded(r, [&]() -> int {
		if (!r) return SDFR_ERR_INVALID_ARGUMENT;
		if (!r->have_render) return SDFR_OK;
		SDFR_HIP(hipSetDevice(r->device));
		SDFR_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, r->ev_end, 0));
		return SDFR_OK;
	});
}

int sdfr_create(int device_ordinal, sdfr_renderer **out)
{
	return guarded(nullptr, [&]() -> int {
		if (!out) return SDFR_ERR_INVALID_ARGUMENT;
		*out = nullptr;
		int count = 0;
		if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SDFR_ERR_NO_DEVICE;
		if (device_ordinal < 0 || device_ordinal >= count) return SDFR_ERR_INVALID_ARGUMENT;
		if (hipSetDevice(device_ordinal) != hipSuccess) return SDFR_ERR_HIP;
		sdfr_renderer *r = new sdfr_renderer();
		r->device = device_ordinal;
		if (const char *t = getenv("SDFR_TILE_W_LOG2")) // developer knob: wave tile shape (3 = 8x8 ... 6 = 64x1)
		{
			int v = atoi(t);
			if (v >= 3 && v <= 6) r->tile_w_log2 = v;
		}
		if (const char *t = getenv("SDFR_STEP_SHORTCUTS")) r->step_shortcuts = atoi(t) != 0; // default of sdfr_set_step_shortcuts
		frame_defaults(r->U);
		// start-up camera of the reference (Application.cpp:214-224), aspect of its 1200x800 window
		host::Camera cam;
		cam.SetAspect(1200.f / 800.f);
		cam.SetFOVY(60.f * 3.14159265358979f / 180.f);
		cam.SetRoll(0.f);
		cam.SetEye(host::Vec3(0.f, 2.f, -3.f));
		cam.SetLookat(host::Vec3(0.f, 1.f, 0.f));
		host::Vec3 e, f, rt, tp;
		cam.GetBasis(e, f, rt, tp);
		r->U.eye = V3(e.x, e.y, e.z);
		r->U.front = V3(f.x, f.y, f.z);
		r->U.right = V3(rt.x, rt.y, rt.z);
		r->U.top = V3(tp.x, tp.y, tp.z);
		if (hipMalloc((void **)&r->d_totals, 2 * sizeof(RenderTotals)) != hipSuccess || hipEventCreate(&r->ev_begin) != hipSuccess ||
			hipEventCreate(&r->ev_end) != hipSuccess)
		{
			delete r;
			return SDFR_ERR_HIP;
		}
		for (int i = 0; i < 32; ++i)
		{
			if (i < 3) (void)hipEventCreate(&r->ev_post[i]);
			(void)hipEventCreate(&r->ev_march[i]);
			(void)hipEventCreate(&r->ev_shade[i]);
		}
		*out = r;
		return SDFR_OK;
	});
}

void sdfr_destroy(sdfr_renderer *r)
{
	if (!r) return;
	(void)hipSetDevice(r->device);
	(void)hipStreamSynchronize(r->stream);
	if (r->comm_stream) (void)hipStreamSynchronize(r->comm_stream);
	comm_forget_renderer(r);
	release_second_lane(r); // waits for it, frees its workspace, counters, events and the two internal streams
	free_workspace(r);
	jit_unload(r->jit);
	(void)hipFree(r->d_totals);
	(void)hipFree(r->d_stage);
	(void)hipFree(r->d_pstat);
	(void)hipFree(r->d_wire);
	(void)hipFree(r->d_post_flags);
	if (r->pinned_host) (void)hipHostUnregister(r->pinned_host);
	if (r->comm_stream) (void)hipStreamDestroy(r->comm_stream);
	if (r->ev_strips) (void)hipEventDestroy(r->ev_strips);
	if (r->ev_gathered) (void)hipEventDestroy(r->ev_gathered);
	for (hipEvent_t e : r->ev_xfer)
		if (e) (void)hipEventDestroy(e);
	(void)hipEventDestroy(r->ev_begin);
	(void)hipEventDestroy(r->ev_end);
	for (hipEvent_t e : r->ev_post) (void)hipEventDestroy(e);
	for (int i = 0; i < 32; ++i)
	{
		(void)hipEventDestroy(r->ev_march[i]);
		(void)hipEventDestroy(r->ev_shade[i]);
	}
	delete r;
}

const char *sdfr_last_error(const sdfr_renderer *r) { return r ? r->error.c_str() : "null handle"; }

int sdfr_set_stream(sdfr_renderer *r, void *hip_stream)
{
	return guarded(r, [&]() -> int {
		if (!r) return SDFR_ERR_INVALID_ARGUMENT;
		r->user_stream = (hipStream_t)hip_stream;
		if (r->frames_in_flight == 1) r->stream = r->user_stream; // (two frames in flight run on the handle's own two streams)
		return SDFR_OK;
	});
}

int sdfr_scene_count(void) { return SDFR_PUBLIC_SCENE_COUNT; }
const char *sdfr_scene_name(int index) { return index >= 0 && index < SDFR_PUBLIC_SCENE_COUNT ? scene_name(index) : nullptr; }

// rebuild the variable table like SDFRenderer::initShader (SDFRenderer.cpp:35-47): clear, then
// collect the tags of the driver and of the scene text
static int build_variable_table(sdfr_renderer *r, const std::string &scene_text, host::ShaderVariableManager &vm, std::vector<std::string> &slots)
{
	// scene slots: distinct names of the scene's tags in order of appearance
	std::vector<std::string_view> code, tags;
	host::split_tagged(scene_text, "VAR_", ")", code, tags);
	for (std::string_view t : tags)
	{
		const size_t lb = t.find('(');
		if (lb == std::string_view::npos || lb <= 4) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "malformed VAR_ tag: '" + std::string(t) + "'");
		std::string nm(t.substr(4, lb - 4));
		for (char c : nm)
			if (!isalnum((unsigned char)c) && c != '_') return fail(r, SDFR_ERR_INVALID_ARGUMENT, "malformed VAR_ tag: '" + std::string(t) + "'");
		bool seen = false;
		for (const auto &s : slots) seen = seen || s == nm;
		if (!seen) slots.push_back(nm);
	}
	if (slots.size() > SDFR_MAX_SCENE_VARS) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "too many scene variables");
	try
	{
		if (!vm.parseFile(std::string(k_driver_variables) + " " + scene_text)) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "malformed VAR_ tag");
	}
	catch (const std::exception &) // a value that is not a number
	{
		return fail(r, SDFR_ERR_INVALID_ARGUMENT, "malformed VAR_ tag: value is not a number");
	}
	return SDFR_OK;
}

int sdfr_load_scene(sdfr_renderer *r, const char *name)
{
	return guarded(r, [&]() -> int {
		if (!r || !name) return SDFR_ERR_INVALID_ARGUMENT;
		const int idx = scene_index(name);
		if (idx < 0) return fail(r, SDFR_ERR_UNKNOWN_SCENE, std::string("unknown scene '") + name + "'");
		host::ShaderVariableManager vm;
		std::vector<std::string> slots;
		const int rc = build_variable_table(r, scene_variables(idx), vm, slots);
		if (rc != SDFR_OK) return rc;
		r->vars = vm;
		r->scene_var_slots = slots;
		r->scene = idx;
		return SDFR_OK;
	});
}

int sdfr_check_scene_source(const char *source, const char *arch, char *log, size_t log_bytes)
{
	return guarded(nullptr, [&]() -> int {
		if (!source) return SDFR_ERR_INVALID_ARGUMENT;
		if (log && log_bytes) log[0] = 0;
		sdfr_renderer scratch; // only its error string is used
		host::ShaderVariableManager vm;
		std::vector<std::string> slots;
		std::string err;
		int rc = build_variable_table(&scratch, source, vm, slots);
		if (rc != SDFR_OK) err = scratch.error;
		std::vector<char> code;
		if (rc == SDFR_OK && !jit_compile_code(arch && arch[0] ? arch : "gfx950", "scene", source, slots, code, err)) rc = SDFR_ERR_COMPILE;
		if (rc != SDFR_OK && log && log_bytes) snprintf(log, log_bytes, "%s", err.c_str());
		return rc;
	});
}

// ---- scenes in the reference's own dialect (sdfr_hlsl.h / sdfr_hlsl.cpp) ------------------------------------
int sdfr_translate_scene_hlsl(const char *hlsl_source, char *out, size_t out_bytes)
{
	return guarded(nullptr, [&]() -> int {
		if (!hlsl_source) return SDFR_ERR_INVALID_ARGUMENT;
		const std::string text = hlsl_scene_source(hlsl_source);
		if (out && out_bytes) snprintf(out, out_bytes, "%s", text.c_str());
		return (int)text.size() + 1;
	});
}

int sdfr_check_scene_hlsl(const char *hlsl_source, const char *arch, char *log, size_t log_bytes)
{
	return guarded(nullptr, [&]() -> int {
		if (!hlsl_source) return SDFR_ERR_INVALID_ARGUMENT;
		// the variable table comes from the tags of the ORIGINAL text (ShaderUtil.cpp:122-191); the generated class reads them
		// through the same VAR_<name>(...) macros as any run-time scene
		if (log && log_bytes) log[0] = 0;
		sdfr_renderer scratch;
		host::ShaderVariableManager vm;
		std::vector<std::string> slots;
		std::string err;
		int rc = build_variable_table(&scratch, hlsl_source, vm, slots);
		if (rc != SDFR_OK) err = scratch.error;
		std::vector<char> code;
		if (rc == SDFR_OK && !jit_compile_code(arch && arch[0] ? arch : "gfx950", "scene", hlsl_scene_source(hlsl_source), slots, code, err)) rc = SDFR_ERR_COMPILE;
		if (rc != SDFR_OK && log && log_bytes) snprintf(log, log_bytes, "%s", err.c_str());
		return rc;
	});
}

int sdfr_load_scene_hlsl(sdfr_renderer *r, const char *name, const char *hlsl_source)
{
	return guarded(r, [&]() -> int {
		if (!r || !name || !hlsl_source) return SDFR_ERR_INVALID_ARGUMENT;
		SDFR_HIP(hipSetDevice(r->device));
		host::ShaderVariableManager vm;
		std::vector<std::string> slots;
		const int rc = build_variable_table(r, hlsl_source, vm, slots);
		if (rc != SDFR_OK) return rc;
		JitScene js;
		std::string err;
		if (!jit_compile(r->device, name, hlsl_scene_source(hlsl_source), slots, js, err)) return fail(r, SDFR_ERR_COMPILE, err);
		SDFR_HIP(hipStreamSynchronize(r->stream));
		if (r->frames_in_flight == 2) SDFR_HIP(hipStreamSynchronize(r->other.stream)); // the frame before may still run the old module
		jit_unload(r->jit);
		r->jit = js;
		r->vars = vm;
		r->scene_var_slots = slots;
		r->scene = SDFR_SCENE_COUNT;
		return SDFR_OK;
	});
}

int sdfr_load_scene_source(sdfr_renderer *r, const char *name, const char *source)
{
	return guarded(r, [&]() -> int {
		if (!r || !name || !source) return SDFR_ERR_INVALID_ARGUMENT;
		SDFR_HIP(hipSetDevice(r->device));
		host::ShaderVariableManager vm;
		std::vector<std::string> slots;
		const int rc = build_variable_table(r, source, vm, slots);
		if (rc != SDFR_OK) return rc;
		JitScene js;
		std::string err;
		// like the reference, a scene that fails to compile leaves the previous one in place
		// (SceneManager.cpp:118-127 keeps the old shader and shows the compiler's message)
		if (!jit_compile(r->device, name, source, slots, js, err)) return fail(r, SDFR_ERR_COMPILE, err);
		SDFR_HIP(hipStreamSynchronize(r->stream));
		if (r->frames_in_flight == 2) SDFR_HIP(hipStreamSynchronize(r->other.stream));
		jit_unload(r->jit);
		r->jit = js;
		r->vars = vm;
		r->scene_var_slots = slots;
		r->scene = SDFR_SCENE_COUNT;
		return SDFR_OK;
	});
}

const char *sdfr_current_scene(const sdfr_renderer *r)
{
	if (!r || r->scene < 0) return nullptr;
	return r->scene == SDFR_SCENE_COUNT ? r->jit.name.c_str() : scene_name(r->scene);
}

int sdfr_var_count(const sdfr_renderer *r) { return r ? (int)r->vars.getVariables().size() : SDFR_ERR_INVALID_ARGUMENT; }

int sdfr_var_info(const sdfr_renderer *r, int index, sdfr_variable *out)
{
	return guarded(r, [&]() -> int {
		if (!r || !out) return SDFR_ERR_INVALID_ARGUMENT;
		const auto &m = r->vars.getVariables();
		if (index < 0 || index >= (int)m.size()) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "variable index out of range");
		auto it = m.begin();
		std::advance(it, index);
		memset(out, 0, sizeof *out);
		snprintf(out->name, sizeof out->name, "%s", it->first.c_str());
		out->minval = it->second.minval;
		out->maxval = it->second.maxval;
		out->start = it->second.start;
		out->step = it->second.step;
		out->value = it->second.value;
		return SDFR_OK;
	});
}

int sdfr_var_set(sdfr_renderer *r, const char *name, float value)
{
	return guarded(r, [&]() -> int {
		if (!r || !name) return SDFR_ERR_INVALID_ARGUMENT;
		if (!r->vars.setValue(name, value)) return fail(r, SDFR_ERR_UNKNOWN_VARIABLE, std::string("unknown variable '") + name + "' (ignored)");
		return SDFR_OK;
	});
}

int sdfr_var_get(const sdfr_renderer *r, const char *name, float *out)
{
	return guarded(r, [&]() -> int {
		if (!r || !name || !out) return SDFR_ERR_INVALID_ARGUMENT;
		const auto &m = r->vars.getVariables();
		auto it = m.find(std::string_view(name));
		if (it == m.end()) return fail(r, SDFR_ERR_UNKNOWN_VARIABLE, std::string("unknown variable '") + name + "'");
		*out = it->second.value;
		return SDFR_OK;
	});
}

int sdfr_vars_reset(sdfr_renderer *r)
{
	return guarded(r, [&]() -> int {
		if (!r) return SDFR_ERR_INVALID_ARGUMENT;
		for (auto &kv : r->vars.getVariables()) kv.second.value = kv.second.start;
		return SDFR_OK;
	});
}

int sdfr_set_camera(sdfr_renderer *r, const float eye[3], const float front[3], const float right[3], const float top[3])
{
	return guarded(r, [&]() -> int {
		if (!r || !eye || !front || !right || !top) return SDFR_ERR_INVALID_ARGUMENT;
		r->U.eye = V3(eye[0], eye[1], eye[2]);
		r->U.front = V3(front[0], front[1], front[2]);
		r->U.right = V3(right[0], right[1], right[2]);
		r->U.top = V3(top[0], top[1], top[2]);
		return SDFR_OK;
	});
}

static int set_camera_from(sdfr_renderer *r, const host::Camera &cam)
{
	host::Vec3 e, f, rt, tp;
	cam.GetBasis(e, f, rt, tp);
	r->U.eye = V3(e.x, e.y, e.z);
	r->U.front = V3(f.x, f.y, f.z);
	r->U.right = V3(rt.x, rt.y, rt.z);
	r->U.top = V3(tp.x, tp.y, tp.z);
	return SDFR_OK;
}

int sdfr_set_camera_lookat(sdfr_renderer *r, const float eye[3], const float lookat[3], float fovy, float aspect, float roll)
{
	return guarded(r, [&]() -> int {
		if (!r || !eye || !lookat) return SDFR_ERR_INVALID_ARGUMENT;
		host::Camera cam;
		cam.SetAspect(aspect);
		cam.SetFOVY(fovy);
		cam.SetRoll(roll);
		cam.SetEye(host::Vec3(eye[0], eye[1], eye[2]));
		cam.SetLookat(host::Vec3(lookat[0], lookat[1], lookat[2]));
		return set_camera_from(r, cam);
	});
}

int sdfr_set_camera_direction(sdfr_renderer *r, const float eye[3], const float direction[3], float fovy, float aspect, float roll)
{
	return guarded(r, [&]() -> int {
		if (!r || !eye || !direction) return SDFR_ERR_INVALID_ARGUMENT;
		host::Camera cam;
		cam.SetAspect(aspect);
		cam.SetFOVY(fovy);
		cam.SetRoll(roll);
		cam.SetEye(host::Vec3(eye[0], eye[1], eye[2]));
		cam.SetDirection(host::Vec3(direction[0], direction[1], direction[2]));
		return set_camera_from(r, cam);
	});
}

int sdfr_get_camera(const sdfr_renderer *r, float out[12])
{
	return guarded(r, [&]() -> int {
		if (!r || !out) return SDFR_ERR_INVALID_ARGUMENT;
		const vec3 v[4] = {r->U.eye, r->U.front, r->U.right, r->U.top};
		for (int i = 0; i < 4; ++i)
		{
			out[3 * i + 0] = v[i].x;
			out[3 * i + 1] = v[i].y;
			out[3 * i + 2] = v[i].z;
		}
		return SDFR_OK;
	});
}

int sdfr_set_time(sdfr_renderer *r, float stime)
{
	return guarded(r, [&]() -> int {
		if (!r) return SDFR_ERR_INVALID_ARGUMENT;
		r->U.stime = stime;
		return SDFR_OK;
	});
}

int sdfr_get_limits(const sdfr_renderer *r, sdfr_limits *out)
{
	return guarded(r, [&]() -> int {
		if (!r || !out) return SDFR_ERR_INVALID_ARGUMENT;
		out->iter_count = r->U.iter_count;
		out->bounce_count = r->U.bounce_count;
		out->ray_count = r->U.ray_count;
		out->light_count = r->U.light_count;
		out->range = r->U.range;
		out->max_cost_default = (int)r->U.max_cost_default;
		out->extension_lights = r->U.extension_lights;
		out->extension_marble_reflection = r->U.extension_marble_reflection;
		out->dist_eps = r->U.dist_eps;
		out->grad_eps = r->U.grad_eps;
		out->reflect_eps = r->U.reflect_eps;
		out->refract_eps = r->U.refract_eps;
		out->shadow_eps = r->U.shadow_eps;
		return SDFR_OK;
	});
}

int sdfr_set_limits(sdfr_renderer *r, const sdfr_limits *l)
{
	return guarded(r, [&]() -> int {
		if (!r || !l) return SDFR_ERR_INVALID_ARGUMENT;
		if (l->iter_count < 1 || l->iter_count > 0xffffff || l->bounce_count < 0 || l->bounce_count > 16 || l->ray_count < 1 ||
			l->ray_count > SDFR_MAX_RAYS || l->light_count < 0 || l->light_count > SDFR_MAX_LIGHTS || l->max_cost_default < 0 ||
			l->max_cost_default > 250 || !(l->range == l->range) || l->extension_lights < 0 || l->extension_lights > SDFR_MAX_LIGHTS - 1 ||
			!(l->extension_marble_reflection >= 0.f && l->extension_marble_reflection <= 1.f))
			return fail(r, SDFR_ERR_INVALID_ARGUMENT, "limits out of range");
		if (!(l->dist_eps > 0.f && l->dist_eps <= SDFR_MAX_DIST_EPS) || !(l->grad_eps > 0.f && l->grad_eps <= 1.f) || !(l->reflect_eps >= 0.f && l->reflect_eps <= 1.f) ||
			!(l->refract_eps >= 0.f && l->refract_eps <= 1.f) || !(l->shadow_eps >= 0.f && l->shadow_eps <= 1.f))
			return fail(r, SDFR_ERR_INVALID_ARGUMENT, "epsilons out of range (0 < dist_eps <= 1e-3, 0 < grad_eps <= 1, 0 <= reflect_eps, refract_eps, shadow_eps <= 1)");
		r->U.iter_count = l->iter_count;
		r->U.bounce_count = l->bounce_count;
		r->U.ray_count = l->ray_count;
		r->U.light_count = l->light_count;
		r->U.range = l->range;
		r->U.max_cost_default = (uint32_t)l->max_cost_default;
		r->U.extension_lights = l->extension_lights;
		r->U.extension_marble_reflection = l->extension_marble_reflection;
		r->U.dist_eps = l->dist_eps;
		r->U.grad_eps = l->grad_eps;
		r->U.reflect_eps = l->reflect_eps;
		r->U.refract_eps = l->refract_eps;
		r->U.shadow_eps = l->shadow_eps;
		return SDFR_OK;
	});
}

int sdfr_set_profiling(sdfr_renderer *r, int enabled)
{
	return guarded(r, [&]() -> int {
		if (!r) return SDFR_ERR_INVALID_ARGUMENT;
		r->profiling = enabled != 0;
		return SDFR_OK;
	});
}

int sdfr_set_launch_mode(sdfr_renderer *r, int mode)
{
	return guarded(r, [&]() -> int {
		if (!r || mode < SDFR_LAUNCH_AUTO || mode > SDFR_LAUNCH_PERSISTENT) return SDFR_ERR_INVALID_ARGUMENT;
		r->launch_mode = mode;
		return SDFR_OK;
	});
}

int sdfr_set_step_shortcuts(sdfr_renderer *r, int enabled)
{
	return guarded(r, [&]() -> int {
		if (!r) return SDFR_ERR_INVALID_ARGUMENT;
		r->step_shortcuts = enabled != 0;
		return SDFR_OK;
	});
}

int sdfr_set_schedule(sdfr_renderer *r, int schedule)
{
	return guarded(r, [&]() -> int {
		if (!r || (schedule != SDFR_SCHEDULE_WAVEFRONT && schedule != SDFR_SCHEDULE_PIXEL)) return SDFR_ERR_INVALID_ARGUMENT;
		r->schedule = schedule;
		return SDFR_OK;
	});
}

int64_t sdfr_strip_buffer_bytes(int width, int height, int world, int format) { return sdfr_strip_buffer_bytes_split(width, height, world, format, 0, 1); }

int64_t sdfr_strip_buffer_bytes_split(int width, int height, int world, int format, int priv_count, int priv_period)
{
	const int64_t n = sdfr_strip_buffer_pixels_split(width, height, world, priv_count, priv_period);
	if (n < 0 || !is_wire_format(format)) return -1;
	return (int64_t)image_bytes((size_t)n, format);
}

int64_t sdfr_strip_buffer_pixels(int width, int height, int world)
{
	return sdfr_strip_buffer_pixels_split(width, height, world, 0, 1);
}

int64_t sdfr_strip_buffer_pixels_split(int width, int height, int world, int priv_count, int priv_period)
{
	if (width < 1 || height < 1 || world < 1 || priv_count < 0 || priv_period < 1 || priv_count >= priv_period) return 0;
	const int64_t strips = ((int64_t)height + SDFR_STRIP_ROWS - 1) / SDFR_STRIP_ROWS;
	const int64_t shared = strips - (int64_t)private_strip_count((uint32_t)strips, priv_count, priv_period);
	return ((shared + world - 1) / world) * SDFR_STRIP_ROWS * (int64_t)width;
}

int sdfr_set_strip_split(sdfr_renderer *r, int priv_count, int priv_period)
{
	return guarded(r, [&]() -> int {
		if (!r || priv_count < 0 || priv_period < 1 || priv_count >= priv_period || priv_period > 4096) return SDFR_ERR_INVALID_ARGUMENT;
		r->priv_count = priv_count;
		r->priv_period = priv_period;
		return SDFR_OK;
	});
}

// latch the variable values into the frame uniforms (the reference uploads them every frame,
// SDFRenderer.cpp:75-78) and derive the per-frame constants
static int latch_frame(sdfr_renderer *r, int width, int height)
{
	if (r->scene < 0) return fail(r, SDFR_ERR_NO_SCENE, "no scene loaded");
	if (width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 30) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad frame size");
	FrameU &U = r->U;
	U.width = width;
	U.height = height;
	const auto &m = r->vars.getVariables();
	auto val = [&](const char *n) { auto it = m.find(std::string_view(n)); return it != m.end() ? it->second.value : 0.f; };
	U.debug_nx = val("debug_nx");
	U.debug_ny = val("debug_ny");
	U.debug_nz = val("debug_nz");
	U.debug_scale = val("debug_scale");
	U.debug_x = val("debug_x");
	U.debug_y = val("debug_y");
	U.debug_z = val("debug_z");
	U.show_objects = val("show_objects");
	for (int i = 0; i < SDFR_MAX_SCENE_VARS; ++i) U.scene_var[i] = 0.f;
	for (size_t k = 0; k < r->scene_var_slots.size(); ++k) U.scene_var[k] = val(r->scene_var_slots[k].c_str());
	U.step_shortcuts = r->step_shortcuts ? 1 : 0;
	frame_derive(U, r->scene);
	if (r->scene == SDFR_SCENE_COUNT) SDFR_HIP(jit_prepare(r->jit, U, r->stream));
	return SDFR_OK;
}

} // extern "C"

int sdfr::render_impl(sdfr_renderer *r, int width, int height, int rank, int world, void *out, int format, int out_on_host, uint32_t *pixel_stats,
	RenderMode mode, RenderTotals *totals)
{
	if (r && !totals) totals = r->d_totals;
	if (!r || !out) return SDFR_ERR_INVALID_ARGUMENT;
	const bool strips = mode == RENDER_STRIPS;
	if (format != SDFR_RGBA32F && format != SDFR_RGBA16F && !(strips && is_wire_format(format)))
		return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad format");
	if (mode == RENDER_PRIVATE && r->priv_count == 0) return SDFR_OK; // no private strips: nothing to render
	if (world < 1 || rank < 0 || rank >= world) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad rank/world");
	SDFR_HIP(hipSetDevice(r->device));
	const auto t_setup = std::chrono::steady_clock::now();
	int rc = latch_frame(r, width, height);
	if (rc != SDFR_OK) return rc;
	r->ms_setup = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_setup).count();

	RowMap rm;
	rm.rank = rank;
	rm.world = world;
	rm.tile_w_log2 = r->tile_w_log2 ? r->tile_w_log2 : scene_tile_w_log2(r->scene);
	rm.priv_count = mode == RENDER_FULL ? 0 : r->priv_count;
	rm.priv_period = mode == RENDER_FULL ? 1 : r->priv_period;
	rm.direct = mode == RENDER_PRIVATE ? 1 : 0;
	row_map_tiles(rm, width);
	rm.unit_log2 = rm.units_x = rm.units_x_magic = rm.units = 0u; // the launcher of a persistent launch decides (row_map_units)
	rm.retire_after = 0u;
	rm.feedback_key = 0u;
	const uint32_t frame_strips = (uint32_t)((height + SDFR_STRIP_ROWS - 1) / SDFR_STRIP_ROWS);
	if (mode == RENDER_FULL)
		rm.local_rows = height;
	else if (mode == RENDER_STRIPS) // a strip buffer keeps whole strips (rows past the frame stay zero)
		rm.local_rows = (int)(sdfr_strip_buffer_pixels_split(width, height, world, rm.priv_count, rm.priv_period) / width);
	else
		rm.local_rows = (int)(private_strip_count(frame_strips, rm.priv_count, rm.priv_period) * SDFR_STRIP_ROWS);
	// pixels the output (and the per-pixel workspace, which is indexed like the output) spans
	const size_t local_pixels = mode == RENDER_PRIVATE ? (size_t)width * height : (size_t)rm.local_rows * width;
	const size_t out_bytes = image_bytes(local_pixels, format);

	void *d_out = out;
	uint32_t *d_pstat = pixel_stats;
	if (out_on_host)
	{
		if (r->stage_bytes < out_bytes)
		{
			(void)hipFree(r->d_stage);
			r->d_stage = nullptr;
			r->stage_bytes = 0;
			SDFR_HIP(hipMalloc(&r->d_stage, out_bytes));
			r->stage_bytes = out_bytes;
		}
		d_out = r->d_stage;
		if (pixel_stats)
		{
			if (r->pstat_bytes < local_pixels * 12)
			{
				(void)hipFree(r->d_pstat);
				r->d_pstat = nullptr;
				r->pstat_bytes = 0;
				SDFR_HIP(hipMalloc((void **)&r->d_pstat, local_pixels * 12));
				r->pstat_bytes = local_pixels * 12;
			}
			d_pstat = r->d_pstat;
		}
	}
	if (strips)
	{
		// rows of the buffer past the end of the frame are never written: define them (only the ranks
		// whose last strip is missing or cut short have any)
		const int last_local_strip = rm.local_rows / SDFR_STRIP_ROWS - 1;
		const long long last_row_end = last_local_strip < 0 ? 0 : ((long long)strip_local_to_global(rm, (uint32_t)last_local_strip) + 1) * SDFR_STRIP_ROWS;
		if (last_row_end > height && out_bytes) SDFR_HIP(hipMemsetAsync(d_out, 0, out_bytes, r->stream));
	}

	const bool pixel_schedule = r->scene == SDFR_SCENE_COUNT || r->schedule == SDFR_SCHEDULE_PIXEL;
	if (!pixel_schedule) SDFR_HIP(hipMemsetAsync(totals, 0, sizeof(RenderTotals), r->stream)); // the wavefront kernels add to it
	hipError_t e;
	if (rm.local_rows == 0) return SDFR_OK; // e.g. every strip of a small frame is private
	{
		size_t need = (size_t)launch_capacity_items(width, rm);
		if (need < local_pixels) need = local_pixels; // private strips index the workspace by image position
		rc = ensure_workspace(r, need, !pixel_schedule);
	}
	if (rc != SDFR_OK) return rc;
	if (!r->caller_times) SDFR_HIP(hipEventRecord(r->ev_begin, r->stream));
	if (r->scene == SDFR_SCENE_COUNT) // scenes compiled at run time exist for the PIXEL schedule only
	{
		e = jit_launch_pixel(r->jit, r->U, rm, d_out, format, d_pstat, totals, r->ws, r->stream, r->launch_mode);
		r->last_wavefront = false;
	}
	else if (r->schedule == SDFR_SCHEDULE_PIXEL)
	{
		e = launch_pixel_schedule(r->scene, r->U, rm, d_out, format, d_pstat, totals, r->ws, r->stream, r->launch_mode);
		r->last_wavefront = false;
	}
	else
	{
		e = launch_wavefront_schedule(r->scene, r->U, rm, d_out, format, d_pstat, totals, r->ws, r->stream, r->profiling ? r->ev_march : nullptr,
			r->profiling ? r->ev_shade : nullptr, &r->last_rounds);
		r->last_wavefront = true;
		r->last_profiled = r->profiling;
	}
	if (e != hipSuccess) return hip_fail(r, e, "kernel launch");
	if (!r->caller_times)
	{
		SDFR_HIP(hipEventRecord(r->ev_end, r->stream));
		r->totals_parts = 1;
	}
	r->have_render = true;

	if (out_on_host)
	{
		SDFR_HIP(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, r->stream));
		if (pixel_stats) SDFR_HIP(hipMemcpyAsync(pixel_stats, d_pstat, local_pixels * 12, hipMemcpyDeviceToHost, r->stream));
		SDFR_HIP(hipStreamSynchronize(r->stream));
	}
	return SDFR_OK;
}

extern "C" {

int sdfr_render(sdfr_renderer *r, int width, int height, void *out, int format, int out_on_host, uint32_t *pixel_stats)
{
	return guarded(r, [&]() -> int {
		if (r && r->frames_in_flight == 2 && out && width > 0 && height > 0)
		{
			// the lane of the frame before last takes this one; its stream orders it behind that frame (same workspace)
			SDFR_HIP(hipSetDevice(r->device));
			swap_lanes(r);
			const char *lo = out_on_host ? nullptr : (const char *)out;
			const char *hi = lo ? lo + image_bytes((size_t)width * height, format) : nullptr;
			// the frame still in flight on the other lane writes [other.out_lo, other.out_hi): the same memory twice in a row is
			// a hazard between the two streams
			if (lo && r->other.out_lo && lo < r->other.out_hi && r->other.out_lo < hi && r->other.have_render)
				SDFR_HIP(hipStreamWaitEvent(r->stream, r->other.ev_end, 0));
			r->out_lo = lo;
			r->out_hi = hi;
		}
		return render_impl(r, width, height, 0, 1, out, format, out_on_host, pixel_stats, RENDER_FULL);
	});
}

int sdfr_render_strips(sdfr_renderer *r, int width, int height, int rank, int world, void *out_compact, int format)
{
	return guarded(r, [&]() -> int {
		return render_impl(r, width, height, rank, world, out_compact, format, 0, nullptr, RENDER_STRIPS);
	});
}

int sdfr_render_private_strips(sdfr_renderer *r, int width, int height, void *out_image, int format)
{
	return guarded(r, [&]() -> int {
		return render_impl(r, width, height, 0, 1, out_image, format, 0, nullptr, RENDER_PRIVATE);
	});
}

int sdfr_assemble_strips(sdfr_renderer *r, int width, int height, int world, const void *gathered, void *out_image, int format)
{
	return guarded(r, [&]() -> int {
		if (!r || !gathered || !out_image || width < 1 || height < 1 || world < 1) return SDFR_ERR_INVALID_ARGUMENT;
		if (!is_wire_format(format)) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad format");
		SDFR_HIP(hipSetDevice(r->device));
		hipError_t e = launch_assemble_strips(width, height, world, gathered, out_image, format, r->priv_count, r->priv_period, r->stream);
		if (e != hipSuccess) return hip_fail(r, e, "assemble launch");
		return SDFR_OK;
	});
}

int sdfr_postprocess(sdfr_renderer *r, int width, int height, const void *scene_rgba16f, void *bloom_scratch_rgba16f, void *out_rgba8)
{
	return guarded(r, [&]() -> int {
		if (!r || !scene_rgba16f || !bloom_scratch_rgba16f || !out_rgba8) return SDFR_ERR_INVALID_ARGUMENT;
		if (width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 30) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad frame size");
		SDFR_HIP(hipSetDevice(r->device));
		const size_t flag_bytes = postprocess_flag_bytes(width, height);
		if (r->post_flag_bytes < flag_bytes)
		{
			SDFR_HIP(hipStreamSynchronize(r->stream)); // a postprocess still in flight reads the old one
			(void)hipFree(r->d_post_flags);
			r->d_post_flags = nullptr;
			r->post_flag_bytes = 0;
			SDFR_HIP(hipMalloc((void **)&r->d_post_flags, flag_bytes));
			r->post_flag_bytes = flag_bytes;
		}
		SDFR_HIP(hipEventRecord(r->ev_post[0], r->stream));
		hipError_t e = launch_postprocess(width, height, scene_rgba16f, bloom_scratch_rgba16f, out_rgba8, r->d_post_flags, r->stream, r->ev_post[1]);
		if (e != hipSuccess) return hip_fail(r, e, "postprocess launch");
		SDFR_HIP(hipEventRecord(r->ev_post[2], r->stream));
		r->have_post = true;
		return SDFR_OK;
	});
}

int sdfr_selftest_exception(sdfr_renderer *r, int what)
{
	return guarded(r, [&]() -> int {
		if (what == 0) throw std::runtime_error("sdfr_selftest_exception: thrown on purpose");
		if (what == 1) throw std::bad_alloc();
		if (what == 2) throw 42;
		return SDFR_OK;
	});
}

int sdfr_selftest_math(sdfr_renderer *r, int what, float constant, uint64_t *mismatches)
{
	return guarded(r, [&]() -> int {
		if (!r || !mismatches || what < 0 || what > 5) return SDFR_ERR_INVALID_ARGUMENT;
		if (what >= 1 && !(constant != 0.f && constant == constant)) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad divisor");
		SDFR_HIP(hipSetDevice(r->device));
		unsigned long long *d = nullptr;
		SDFR_HIP(hipMalloc((void **)&d, sizeof(unsigned long long)));
		hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), r->stream);
		if (e == hipSuccess) e = launch_selftest_math(what, constant, d, r->stream);
		unsigned long long h = 0;
		if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, r->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
		(void)hipFree(d);
		if (e != hipSuccess) return hip_fail(r, e, "selftest");
		*mismatches = h;
		return SDFR_OK;
	});
}

#ifdef SDFR_WAVE_TRACE
// developer build only (tools/wave_trace.py): the per-block records of the last pixel-schedule launch
int sdfr_debug_read_partials(sdfr_renderer *r, void *host, size_t records)
{
	return guarded(r, [&]() -> int {
		if (!r || !host) return SDFR_ERR_INVALID_ARGUMENT;
		SDFR_HIP(hipStreamSynchronize(r->stream));
		if (records > r->ws.capacity / 64 + 1) records = r->ws.capacity / 64 + 1;
		SDFR_HIP(hipMemcpy(host, r->ws.partials, records * sizeof(RenderTotals), hipMemcpyDeviceToHost));
		SDFR_HIP(hipMemset(r->ws.partials, 0, records * sizeof(RenderTotals))); // records the next launch does not write read as empty
		return SDFR_OK;
	});
}
#endif

int sdfr_register_host_target(sdfr_renderer *r, void *host_image, size_t bytes)
{
	return guarded(r, [&]() -> int {
		if (!r || (host_image && bytes == 0)) return SDFR_ERR_INVALID_ARGUMENT;
		SDFR_HIP(hipSetDevice(r->device));
		SDFR_HIP(hipStreamSynchronize(r->stream));
		if (r->pinned_host) (void)hipHostUnregister(r->pinned_host);
		r->pinned_host = nullptr;
		r->pinned_bytes = 0;
		if (!host_image) return SDFR_OK;
		SDFR_HIP(hipHostRegister(host_image, bytes, hipHostRegisterDefault));
		r->pinned_host = host_image;
		r->pinned_bytes = bytes;
		return SDFR_OK;
	});
}

int sdfr_sync(sdfr_renderer *r)
{
	return guarded(r, [&]() -> int {
		if (!r) return SDFR_ERR_INVALID_ARGUMENT;
		SDFR_HIP(hipStreamSynchronize(r->stream));
		if (r->frames_in_flight == 2) SDFR_HIP(hipStreamSynchronize(r->other.stream));
		return SDFR_OK;
	});
}

int sdfr_get_stats(sdfr_renderer *r, sdfr_stats *out)
{
	return guarded(r, [&]() -> int {
		if (!r || !out) return SDFR_ERR_INVALID_ARGUMENT;
		memset(out, 0, sizeof *out);
		if (!r->have_render) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "nothing rendered yet");
		SDFR_HIP(hipEventSynchronize(r->ev_end));
		float ms = 0.f;
		SDFR_HIP(hipEventElapsedTime(&ms, r->ev_begin, r->ev_end));
		out->ms_gpu = ms;
		RenderTotals t[2];
		SDFR_HIP(hipMemcpy(t, r->d_totals, sizeof t, hipMemcpyDeviceToHost));
		for (int k = 0; k < r->totals_parts; ++k)
		{
			out->pixels += t[k].pixels;
			out->rays += t[k].rays;
			out->march_evals += t[k].march_evals;
			out->hits += t[k].hits;
		}
		if (r->last_wavefront)
		{
			for (int i = 0; r->last_profiled && i < r->last_rounds && i < 16; ++i)
			{
				float a = 0.f, b = 0.f;
				if (hipEventElapsedTime(&a, r->ev_march[2 * i], r->ev_march[2 * i + 1]) == hipSuccess) out->ms_march += a;
				if (hipEventElapsedTime(&b, r->ev_shade[2 * i], r->ev_shade[2 * i + 1]) == hipSuccess) out->ms_shade += b;
			}
			out->march_launches = (uint32_t)r->last_rounds;
			out->shade_launches = (uint32_t)r->last_rounds;
		}
		else
		{
			out->march_launches = 1;
		}
		return SDFR_OK;
	});
}

int sdfr_get_timings(sdfr_renderer *r, sdfr_timing *out, int capacity)
{
	return guarded(r, [&]() -> int {
		if (!r || (!out && capacity > 0) || capacity < 0) return SDFR_ERR_INVALID_ARGUMENT;
		int n = 0;
		auto put = [&](const char *name, double ms) {
			if (n < capacity)
			{
				memset(&out[n], 0, sizeof out[n]);
				snprintf(out[n].name, sizeof out[n].name, "%s", name);
				out[n].ms = ms;
			}
			++n;
		};
		if (r->have_render)
		{
			SDFR_HIP(hipEventSynchronize(r->ev_end));
			float ms = 0.f;
			SDFR_HIP(hipEventElapsedTime(&ms, r->ev_begin, r->ev_end));
			put("setup", r->ms_setup);
			put("draw", ms);
			for (int i = 0; r->last_wavefront && r->last_profiled && i < r->last_rounds && i < 16; ++i)
			{
				float a = 0.f, b = 0.f;
				char nm[32];
				if (hipEventElapsedTime(&a, r->ev_march[2 * i], r->ev_march[2 * i + 1]) != hipSuccess) continue;
				if (hipEventElapsedTime(&b, r->ev_shade[2 * i], r->ev_shade[2 * i + 1]) != hipSuccess) continue;
				snprintf(nm, sizeof nm, "draw: march %d", i);
				put(nm, a);
				snprintf(nm, sizeof nm, "draw: shade %d", i);
				put(nm, b);
			}
		}
		if (r->have_render && r->have_xfer)
		{
			// the last sdfr_render_gather's transfer on the comm stream: rank 0 receives world - 1 messages at once (one per link),
			// a peer sends one; the bytes ride in the name so that a caller can turn the time into a rate
			SDFR_HIP(hipEventSynchronize(r->ev_xfer[1]));
			float t = 0.f;
			SDFR_HIP(hipEventElapsedTime(&t, r->ev_xfer[0], r->ev_xfer[1]));
			char nm[32];
			snprintf(nm, sizeof nm, "gather transfer %zu B", r->xfer_bytes);
			put(nm, t);
		}
		if (r->have_post)
		{
			SDFR_HIP(hipEventSynchronize(r->ev_post[2]));
			float a = 0.f, b = 0.f;
			SDFR_HIP(hipEventElapsedTime(&a, r->ev_post[0], r->ev_post[1]));
			SDFR_HIP(hipEventElapsedTime(&b, r->ev_post[1], r->ev_post[2]));
			put("Bloom 1", a);
			put("Bloom 2 + HDR", b); // the vertical blur and the tone map are one kernel here
		}
		return n;
	});
}

} // extern "C"
