// sdfr_hostframe.h -- host side of the frame uniforms: reference defaults, the values
// derived once per frame, and the per-scene prepare() dispatch.
#pragma once
#include "sdfr_perpixel.h"

namespace sdfr {

// reference limits (pshader_sdf.hlsl:60-64,350) and driver-variable defaults (F10 rules
// applied to pshader_sdf.hlsl:88-108,142)
inline void frame_defaults(FrameU &U)
{
	memset(&U, 0, sizeof U);
	U.iter_count = 100;
	U.bounce_count = 16;
	U.ray_count = 8;
	U.light_count = 8;
	U.range = 100.f;
	U.max_cost_default = 7;
	U.debug_scale = 0.2f;
	U.show_objects = 1.f;
	U.dist_eps = SDFR_DEFAULT_DIST_EPS; // pshader_sdf.hlsl:31-35
	U.grad_eps = SDFR_DEFAULT_GRAD_EPS;
	U.reflect_eps = SDFR_DEFAULT_REFLECT_EPS;
	U.refract_eps = SDFR_DEFAULT_REFRACT_EPS;
	U.shadow_eps = SDFR_DEFAULT_SHADOW_EPS;
}

inline void frame_derive(FrameU &U, int scene_index)
{
	vec3 n = V3(U.debug_nx, U.debug_ny, U.debug_nz);
	U.debug_plane_on = any3(n) ? 1 : 0;
	U.debug_normal = U.debug_plane_on ? normalize(n) : V3s(0.f);
	U.show_on = (U.show_objects != 0.f) ? 1 : 0;
	U.widthf = (float)U.width;
	U.heightf = (float)U.height;
	U.ddx = 2.f / U.widthf;
	U.ddy = -2.f / U.heightf;
	vec2 sc = sincos1(-U.stime * 0.025f);
	U.sky_s = sc.x;
	U.sky_c = sc.y;
	for (int i = 0; i < SDFR_SCENE_UNIFORMS; ++i)
		U.su[i] = 0.f;
	// extension lights: phi_i = stime * 0.25 + i * (2 pi / 7); (5 cos phi, 3, 5 sin phi); the hue
	// wheel of sdf_scene_light_shadows.hlsl:5-11 at h = i / 7, unit brightness, times 0.5
	for (int i = 1; i < SDFR_MAX_LIGHTS; ++i)
	{
		float *L = U.ext_light[i - 1];
		for (int k = 0; k < 6; ++k) L[k] = 0.f;
		if (i > U.extension_lights) continue;
		const float phi = U.stime * 0.25f + (float)i * (6.28318530718f / 7.f);
		vec3 c = hsv_to_rgb(V3((float)i / 7.f, 1.f, 1.f));
		c = c / rgb_to_brightness(c);
		c = c * 0.5f;
		L[0] = cos1(phi) * 5.f;
		L[1] = 3.f;
		L[2] = sin1(phi) * 5.f;
		L[3] = c.x;
		L[4] = c.y;
		L[5] = c.z;
	}
	switch (scene_index)
	{
#define SDFR_PREP(I, S) case I: S::prepare(U); break;
		SDFR_FOR_EACH_SCENE(SDFR_PREP)
#undef SDFR_PREP
	default: break;
	}
}

inline const char *scene_name(int i)
{
	switch (i)
	{
#define SDFR_NAME(I, S) case I: return S::name();
		SDFR_FOR_EACH_SCENE(SDFR_NAME)
#undef SDFR_NAME
	default: return nullptr;
	}
}
inline const char *scene_variables(int i)
{
	switch (i)
	{
#define SDFR_VARS(I, S) case I: return S::variables();
		SDFR_FOR_EACH_SCENE(SDFR_VARS)
#undef SDFR_VARS
	default: return "";
	}
}
inline int scene_index(const char *name)
{
	for (int i = 0; i < SDFR_SCENE_COUNT; ++i)
		if (strcmp(scene_name(i), name) == 0) return i;
	return -1;
}

} // namespace sdfr
