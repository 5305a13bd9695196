// sdfr_jit.h -- scenes compiled at run time with hiprtc (sdfr_jit.cpp).
#pragma once
#include "sdfr_kernels.h"

#include <string>
#include <vector>

namespace sdfr {

struct JitScene
{
	std::string name;
	hipModule_t module = nullptr;
	hipFunction_t prepare = nullptr, pixel = nullptr, pixel_debug = nullptr;
	FrameU *d_frame = nullptr; // device copy of the frame uniforms for Scene::prepare
};

// the translation unit compiled for a scene: variable macros, the scene text, the kernels
std::string jit_translation_unit(const std::string &scene_source, const std::vector<std::string> &var_slots);

// compile only (no device needed): code object for `arch_name` ("gfx950")
bool jit_compile_code(const std::string &arch_name, const std::string &name, const std::string &scene_source, const std::vector<std::string> &var_slots,
	std::vector<char> &code, std::string &error);
// compile + load; on failure `error` carries the compiler log
bool jit_compile(int device, const std::string &name, const std::string &scene_source, const std::vector<std::string> &var_slots, JitScene &out,
	std::string &error);
void jit_unload(JitScene &js);

// runs Scene::prepare(U) on the device and brings the frame uniforms back (synchronises `stream`)
hipError_t jit_prepare(const JitScene &js, FrameU &U, hipStream_t stream);

hipError_t jit_launch_pixel(const JitScene &js, const FrameU &U, const RowMap &rm, void *out, int format, uint32_t *pixel_stats,
	RenderTotals *totals, const WavefrontWorkspace &ws, hipStream_t stream, int launch_mode = 0);

} // namespace sdfr
