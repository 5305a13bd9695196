// sdfr.hpp -- C++ host mirror of the reference's SDF render stage, header-only over the C ABI
// (include/sdfr.h).  A user of the reference's `class SDFRenderer` (Engine/SDFRenderer.h:17-44)
// and `class Camera` (Engine/Camera.h:5-64, FPS mode) finds the same methods with the same
// meaning here; the D3D plumbing arguments (Graphics&, ShaderIncluder&, FullscreenQuad&,
// GPUProfiler&) are replaced by what this build needs (device ordinal, scene name, target).
#pragma once
#include "sdfr.h"

#include <map>
#include <string>
#include <string_view>

namespace sdfr {

// Engine/ShaderVariable.h:6-12
struct Variable
{
	float minval, maxval, start, step;
	float value; // the current value
};
using VariableMap = std::map<std::string, Variable, std::less<>>;

struct Vector3
{
	float x = 0.f, y = 0.f, z = 0.f;
	Vector3() {}
	Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

// Parameters of the reference's first-person camera; the basis arithmetic
// (Camera.cpp:36-49,156-166) runs inside libsdfr.so.
class Camera
{
public:
	void SetEye(const Vector3 &e) { eye = e; }
	const Vector3 &GetEye() const { return eye; }
	void SetLookat(const Vector3 &l) { target = l; target_is_direction = false; }
	void SetDirection(const Vector3 &d) { target = d; target_is_direction = true; }
	void SetAspect(float a) { aspect = a; }
	float GetAspect() const { return aspect; }
	void SetFOVY(float f) { fovy = f; }
	float GetFOVY() const { return fovy; }
	void SetRoll(float r) { roll = r; }
	float GetRoll() const { return roll; }

	// start-up values of Application.cpp:214-224
	Vector3 eye = Vector3(0.f, 2.f, -3.f), target = Vector3(0.f, 1.f, 0.f);
	bool target_is_direction = false;
	float fovy = 60.f * 3.14159265358979f / 180.f, aspect = 1200.f / 800.f, roll = 0.f;
};

class SDFRenderer
{
public:
	SDFRenderer() = default;
	SDFRenderer(const SDFRenderer &) = delete;
	SDFRenderer &operator=(const SDFRenderer &) = delete;
	~SDFRenderer() { sdfr_destroy(handle); }

	// bool init(Graphics &graphics)
	bool init(int device_ordinal = 0)
	{
		sdfr_destroy(handle);
		handle = nullptr;
		return sdfr_create(device_ordinal, &handle) == SDFR_OK;
	}

	// bool initShader(ShaderIncluder &includer) with the scene substitution of
	// Application::loadScene (Application.cpp:318-322): rebuilds the variable table
	bool initShader(const std::string &scene)
	{
		if (sdfr_load_scene(handle, scene.c_str()) != SDFR_OK) return false;
		return refreshVariables();
	}

	// the same with the scene given as source text, compiled at run time (the reference re-runs
	// initShader when a scene file changes, SceneManager.cpp:102-133); false = does not compile,
	// lastError() has the compiler's messages and the previous scene stays active
	bool initShaderSource(const std::string &name, const std::string &source)
	{
		if (sdfr_load_scene_source(handle, name.c_str(), source.c_str()) != SDFR_OK) return false;
		return refreshVariables();
	}

	// the same with the scene in the reference's own dialect: the text of a scenes/*.hlsl file as the reference compiles it
	// (map / map_normal / map_light / map_background, the OBJECT and MATERIAL macros; sdfr_load_scene_hlsl)
	bool initShaderHlsl(const std::string &name, const std::string &hlsl_text)
	{
		if (sdfr_load_scene_hlsl(handle, name.c_str(), hlsl_text.c_str()) != SDFR_OK) return false;
		return refreshVariables();
	}

	// void setParameters(float stime)
	void setParameters(float stime_) { stime = stime_; }

	// VariableMap &getVariableMap(): values edited in place are latched by the next render,
	// like ShaderVariableManager::updateBuffer (ShaderUtil.cpp:257-267)
	VariableMap &getVariableMap() { return variables; }

	// bool render(FullscreenQuad &quad, GPUProfiler &profiler, Camera &camera):
	// true if it did render something, false otherwise (no valid scene)
	bool render(const Camera &camera, int width, int height, void *target, int format = SDFR_RGBA32F, bool target_on_host = false)
	{
		if (!handle) return false;
		for (const auto &kv : variables) sdfr_var_set(handle, kv.first.c_str(), kv.second.value);
		const float eye[3] = {camera.eye.x, camera.eye.y, camera.eye.z};
		const float tgt[3] = {camera.target.x, camera.target.y, camera.target.z};
		const int rc = camera.target_is_direction ? sdfr_set_camera_direction(handle, eye, tgt, camera.fovy, camera.aspect, camera.roll)
												  : sdfr_set_camera_lookat(handle, eye, tgt, camera.fovy, camera.aspect, camera.roll);
		if (rc != SDFR_OK) return false;
		sdfr_set_time(handle, stime);
		return sdfr_render(handle, width, height, target, format, target_on_host ? 1 : 0, nullptr) == SDFR_OK;
	}

	// two frames in flight inside this renderer (sdfr_set_frames_in_flight): render into two targets in turn, sync() waits for both
	bool setFramesInFlight(int n) { return sdfr_set_frames_in_flight(handle, n) == SDFR_OK; }
	bool sync() { return sdfr_sync(handle) == SDFR_OK; }
	const char *lastError() const { return sdfr_last_error(handle); }
	sdfr_renderer *native() { return handle; }

private:
	bool refreshVariables()
	{
		variables.clear();
		const int n = sdfr_var_count(handle);
		for (int i = 0; i < n; ++i)
		{
			sdfr_variable v;
			if (sdfr_var_info(handle, i, &v) != SDFR_OK) return false;
			variables[v.name] = Variable{v.minval, v.maxval, v.start, v.step, v.value};
		}
		return true;
	}

	sdfr_renderer *handle = nullptr;
	VariableMap variables;
	float stime = 0.f;
};

} // namespace sdfr
