// sdfr_jit.cpp -- scenes compiled at run time.
//
// The reference builds a scene INTO its pixel shader when the scene is selected and again
// whenever its file changes (SceneManager.cpp:102-133, SDFRenderer::initShader
// SDFRenderer.cpp:27-53 with D3DCompile).  The equivalent here: the text of a scene functor
// (the same shape as the ahead-of-time scenes of sdfr_scenes*.h, named `Scene`) is wrapped
// into a translation unit around sdfr_pixel_kernel.h, compiled for the device's architecture
// with hiprtc and loaded as a module.  The pixel kernel body is the one the ahead-of-time
// scenes use, so a scene renders the same bits either way.
//
// VAR_ tags: as in the reference (ShaderUtil.cpp:122-191) the variable table is discovered by
// scanning the source text for `VAR_<name>(key = value, ...)`.  Inside a method that receives
// the frame uniforms as `U`, the tag itself is an expression: the generated prelude defines
// `VAR_<name>(...)` as `(U.scene_var[k])`, k = order of first appearance.
//
// libhiprtc.so is opened on first use: the library has no load-time dependency on it.
#include "sdfr_jit.h"
#include "sdfr_pixel.h"

#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace sdfr {

namespace {

struct Hiprtc
{
	void *lib = nullptr;
	decltype(&hiprtcCreateProgram) create = nullptr;
	decltype(&hiprtcCompileProgram) compile = nullptr;
	decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
	decltype(&hiprtcGetProgramLog) log = nullptr;
	decltype(&hiprtcGetCodeSize) code_size = nullptr;
	decltype(&hiprtcGetCode) code = nullptr;
	decltype(&hiprtcDestroyProgram) destroy = nullptr;
	decltype(&hiprtcGetErrorString) error_string = nullptr;
	bool ok = false;
};

Hiprtc &hiprtc()
{
	static Hiprtc h;
	if (h.lib) return h;
	const char *names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
	for (const char *n : names)
		if ((h.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
	if (!h.lib) return h;
	h.create = (decltype(h.create))dlsym(h.lib, "hiprtcCreateProgram");
	h.compile = (decltype(h.compile))dlsym(h.lib, "hiprtcCompileProgram");
	h.log_size = (decltype(h.log_size))dlsym(h.lib, "hiprtcGetProgramLogSize");
	h.log = (decltype(h.log))dlsym(h.lib, "hiprtcGetProgramLog");
	h.code_size = (decltype(h.code_size))dlsym(h.lib, "hiprtcGetCodeSize");
	h.code = (decltype(h.code))dlsym(h.lib, "hiprtcGetCode");
	h.destroy = (decltype(h.destroy))dlsym(h.lib, "hiprtcDestroyProgram");
	h.error_string = (decltype(h.error_string))dlsym(h.lib, "hiprtcGetErrorString");
	h.ok = h.create && h.compile && h.log_size && h.log && h.code_size && h.code && h.destroy && h.error_string;
	return h;
}

// directory of the kernel headers: $SDFR_JIT_INCLUDE, or csrc/ beside the loaded library
std::string header_dir()
{
	if (const char *e = getenv("SDFR_JIT_INCLUDE")) return e;
	Dl_info info;
	if (dladdr((const void *)&header_dir, &info) && info.dli_fname)
	{
		std::string p = info.dli_fname;
		const size_t slash = p.rfind('/');
		return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/csrc";
	}
	return "csrc";
}

} // namespace

std::string jit_translation_unit(const std::string &scene_source, const std::vector<std::string> &var_slots)
{
	std::string tu;
	tu += "#include \"sdfr_pixel_kernel.h\"\n";
	// a scene that came in the reference's dialect (sdfr_hlsl.cpp) needs the HLSL vocabulary
	if (scene_source.find("hlsl::SceneAdapter") != std::string::npos) tu += "#include \"sdfr_hlsl.h\"\n";
	tu += "namespace sdfr {\n";
	for (size_t k = 0; k < var_slots.size(); ++k)
		tu += "#define VAR_" + var_slots[k] + "(...) (U.scene_var[" + std::to_string(k) + "])\n";
	tu += "#line 1 \"scene\"\n";
	tu += scene_source;
	tu += "\n#line 1 \"sdfr_jit_kernels\"\n";
	tu += "extern \"C\" __global__ void sdfr_jit_prepare(FrameU *U) { if (blockIdx.x == 0 && threadIdx.x == 0) Scene::prepare(*U); }\n";
	tu += "extern \"C\" __global__ SDFR_PIXEL_KERNEL_ATTRS(Scene) void sdfr_jit_pixel(PixelKernelArgs args) { pixel_kernel<Scene, false>(args); }\n";
	tu += "extern \"C\" __global__ SDFR_PIXEL_KERNEL_ATTRS(Scene) void sdfr_jit_pixel_debug(PixelKernelArgs args) { pixel_kernel<Scene, true>(args); }\n";
	tu += "} // namespace sdfr\n";
	return tu;
}

void jit_unload(JitScene &js)
{
	if (js.module) (void)hipModuleUnload(js.module);
	if (js.d_frame) (void)hipFree(js.d_frame);
	js = JitScene();
}

bool jit_compile_code(const std::string &arch_name, const std::string &name, const std::string &scene_source, const std::vector<std::string> &var_slots,
	std::vector<char> &code, std::string &error)
{
	Hiprtc &rtc = hiprtc();
	if (!rtc.ok)
	{
		error = "run-time scene compilation needs libhiprtc.so, which could not be loaded";
		return false;
	}
	const std::string tu = jit_translation_unit(scene_source, var_slots);
	hiprtcProgram prog = nullptr;
	hiprtcResult rc = rtc.create(&prog, tu.c_str(), (name + ".scene.hip").c_str(), 0, nullptr, nullptr);
	if (rc != HIPRTC_SUCCESS)
	{
		error = std::string("hiprtcCreateProgram: ") + rtc.error_string(rc);
		return false;
	}
	// same code generation rules as the ahead-of-time build (buildlib.py): only explicit fma() fuses
	const std::string arch = "--offload-arch=" + arch_name;
	const std::string inc = "-I" + header_dir();
	const std::string block = "-DSDFR_PIXEL_BLOCK=" + std::to_string(pixel_block_threads()); // must match the launch
	// sqrt1 / rcp1 / div_c are exact only on verified domains (sdfr_math.h), which the built-in scenes are
	// checked against and a user's text is not: run-time scenes get the plain IEEE forms unless the text
	// opts in by containing the token SDFR_FAST_EXACT_MATH (say, in a comment)
	const char *math = scene_source.find("SDFR_FAST_EXACT_MATH") != std::string::npos ? "-DSDFR_FAST_EXACT_MATH=1" : "-DSDFR_SAFE_MATH=1";
	const char *opts[] = {arch.c_str(), "-std=c++17", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", inc.c_str(), block.c_str(), math};
	rc = rtc.compile(prog, (int)(sizeof opts / sizeof opts[0]), opts);
	size_t log_bytes = 0;
	rtc.log_size(prog, &log_bytes);
	std::string log(log_bytes, '\0');
	if (log_bytes > 1) rtc.log(prog, log.data());
	if (rc != HIPRTC_SUCCESS)
	{
		error = "scene '" + name + "' does not compile (" + rtc.error_string(rc) + "):\n" + log.c_str();
		rtc.destroy(&prog);
		return false;
	}
	size_t code_bytes = 0;
	rtc.code_size(prog, &code_bytes);
	code.resize(code_bytes);
	rtc.code(prog, code.data());
	rtc.destroy(&prog);
	return true;
}

bool jit_compile(int device, const std::string &name, const std::string &scene_source, const std::vector<std::string> &var_slots, JitScene &out,
	std::string &error)
{
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess)
	{
		error = "hipGetDeviceProperties failed";
		return false;
	}
	std::vector<char> code;
	if (!jit_compile_code(prop.gcnArchName, name, scene_source, var_slots, code, error)) return false;

	JitScene js;
	js.name = name;
	hipError_t e = hipModuleLoadData(&js.module, code.data());
	if (e == hipSuccess) e = hipModuleGetFunction(&js.prepare, js.module, "sdfr_jit_prepare");
	if (e == hipSuccess) e = hipModuleGetFunction(&js.pixel, js.module, "sdfr_jit_pixel");
	if (e == hipSuccess) e = hipModuleGetFunction(&js.pixel_debug, js.module, "sdfr_jit_pixel_debug");
	if (e == hipSuccess) e = hipMalloc((void **)&js.d_frame, sizeof(FrameU));
	if (e != hipSuccess)
	{
		error = std::string("loading the compiled scene failed: ") + hipGetErrorString(e);
		jit_unload(js);
		return false;
	}
	out = js;
	return true;
}

hipError_t jit_prepare(const JitScene &js, FrameU &U, hipStream_t stream)
{
	hipError_t e = hipMemcpyAsync(js.d_frame, &U, sizeof U, hipMemcpyHostToDevice, stream);
	if (e != hipSuccess) return e;
	FrameU *d = js.d_frame;
	void *args[] = {&d};
	e = hipModuleLaunchKernel(js.prepare, 1, 1, 1, 1, 1, 1, 0, stream, args, nullptr);
	if (e != hipSuccess) return e;
	e = hipMemcpyAsync(&U, js.d_frame, sizeof U, hipMemcpyDeviceToHost, stream);
	if (e != hipSuccess) return e;
	return hipStreamSynchronize(stream);
}

hipError_t jit_launch_pixel(const JitScene &js, const FrameU &U, const RowMap &rm, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream, int launch_mode)
{
	uint32_t n_work = launch_work_items(U.width, rm);
	if ((size_t)n_work > ws.capacity) return hipErrorInvalidValue;
	const uint32_t bt = (uint32_t)pixel_block_threads();
	// a persistent launch, as for the scenes compiled ahead of time (run_pixel, sdfr_kernels_group.hip)
	hipFunction_t fn = frame_needs_debug(U) ? js.pixel_debug : js.pixel;
	int per_cu = 0, device = 0;
	if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)bt, 0) != hipSuccess || per_cu < 1) per_cu = 1;
	(void)hipGetDevice(&device);
	const PixelLaunchMode mode = pixel_launch_mode(launch_mode, false); // a run-time scene: one wave per tile unless asked otherwise
	if (mode.blocks_per_cu > 0 && mode.blocks_per_cu < per_cu) per_cu = mode.blocks_per_cu;
	PixelKernelArgs pk;
	pk.U = U;
	pk.rm = rm;
	const uint32_t hand_out_items = n_work; // (tile rows: squares of tiles -- RowMap::unit_log2 -- are a built-in scene's own choice)
	if ((size_t)hand_out_items > ws.capacity) return hipErrorInvalidValue;
	const uint32_t blocks = pixel_launch_blocks(mode, (hand_out_items + bt - 1u) / bt, (uint32_t)(device_cu_count(device) * per_cu));
	pk.rm.retire_after = mode.persistent ? (uint32_t)mode.retire_after : 0u;
	uint32_t name_hash = 2166136261u; // a run-time scene is known by its name
	for (char ch : js.name) name_hash = (name_hash ^ (unsigned char)ch) * 16777619u;
	const uint32_t tiles_x = ((uint32_t)U.width + (1u << rm.tile_w_log2) - 1u) >> rm.tile_w_log2;
	const uint32_t feedback_rows = !mode.persistent ? 0u : pk.rm.unit_log2 ? pk.rm.units : ((n_work + bt - 1u) / bt) / tiles_x;
	pk.rm.feedback_key = mode.persistent ? pixel_feedback_key(0x80000000u | name_hash | (frame_needs_debug(U) ? 1u : 0u), U.width, pk.rm, feedback_rows) : 0u;
	pk.n_work = hand_out_items;
	pk.format = format;
	pk.out = out;
	pk.pixel_stats = pixel_stats;
	pk.partials = ws.partials;
	pk.totals = totals;
	pk.ray_queue = ws.ray_queue;
	pk.cap = ws.capacity;
	pk.tile_cursors = mode.persistent ? ws.tile_cursors : nullptr;
	RenderTotals *partials = ws.partials;
	void *args[] = {&pk};
	const hipError_t e = hipModuleLaunchKernel(fn, blocks, 1, 1, bt, 1, 1, 0, stream, args, nullptr);
	if (e != hipSuccess) return e;
	return launch_reduce_totals(partials, blocks, totals, stream, ws.tile_cursors, feedback_rows, (unsigned long long)n_work, pk.rm.feedback_key);
}

} // namespace sdfr
