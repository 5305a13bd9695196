// sdfr_pixel.h -- the stages of the per-pixel raymarch pipeline (device code,
// host-compilable): primary-ray generation, the 8-slot cost-ordered ray queue, one
// sphere-tracing step, the forward-difference normal, hit shading with secondary-ray
// spawning, and miss shading.
//
// Behavioural contract = the reference's ps_main (Engine/shader/pshader_sdf.hlsl:260-639)
// with its helpers map_geometry (:111-135), map_material (:137-162), grad (:164-177),
// march_ray (:179-220), find_next_ray / find_free_ray (:222-246).  The kernels
// (sdfr_kernels.hip) decide how these stages are scheduled on the machine.
#pragma once
#include "sdfr_frame.h"
#include "sdfr_lib.h"

namespace sdfr {

// ---- pixel -> primary ray (pshader_sdf.hlsl:263-267; pixel centres as D3D rasterises the
// full-screen quad of FullscreenQuad.cpp:52-58, row 0 = top) -------------------------------
struct PixelRay
{
	vec3 dir;
	vec3 right_ray, bottom_ray; // footprint of one pixel per unit distance
};
SDF_HD PixelRay pixel_ray(const FrameU &U, int px, int py)
{
	float sx = ((float)px + 0.5f) / U.widthf * 2.f - 1.f;
	float sy = 1.f - ((float)py + 0.5f) / U.heightf * 2.f;
	vec3 d = U.front + sx * U.right + sy * U.top;
	float invlen = rcp1(length(d));
	PixelRay r;
	r.dir = d * invlen;
	r.right_ray = U.ddx * U.right * invlen;
	r.bottom_ray = U.ddy * U.top * invlen;
	return r;
}
SDF_HD RayRec primary_ray(const FrameU &U, const PixelRay &pr)
{
	RayRec r;
	r.pos = U.eye;
	r.dir = pr.dir;
	r.contrib = V3(1.f, 1.f, 1.f);
	r.shadow_range = 0.f;
	r.bits = 0; // depth 0, outside, no transparency, not a shadow ray
	return r;
}
SDF_HD RayFlags ray_flags(const RayRec &r)
{
	RayFlags f;
	f.has_transparent = ray_has_transparent(r);
	f.is_shadow = ray_is_shadow(r);
	f.last_transparent_pos = f.has_transparent ? r.pos : V3s(0.f);
	return f;
}

// ---- cost-ordered ray queue: depths packed one byte per slot ---------------------------------
// find_next_ray (pshader_sdf.hlsl:222-233): smallest depth, lowest slot wins ties.
SDF_HD int queue_next(uint64_t depths, int slots)
{
	int best = 0;
	uint32_t best_depth = (uint32_t)(depths & 0xffu);
	for (int i = 1; i < slots; ++i)
	{
		uint32_t d = (uint32_t)((depths >> (8 * i)) & 0xffu);
		if (d < best_depth) { best = i; best_depth = d; }
	}
	return best;
}
// find_free_ray (pshader_sdf.hlsl:235-246): first invalid slot
SDF_HD int queue_free(uint64_t depths, int slots)
{
	int i = 0;
	for (; i < slots; ++i)
		if (((depths >> (8 * i)) & 0xffu) == RAY_DEPTH_INVALID) break;
	return i;
}
SDF_HD uint64_t queue_set_depth(uint64_t depths, int slot, uint32_t d)
{
	return (depths & ~(0xffull << (8 * slot))) | ((uint64_t)d << (8 * slot));
}
#define SDFR_QUEUE_EMPTY 0xffffffffffffffffull

// ---- the driver's debug switches (pshader_sdf.hlsl:86-109) -----------------------------------
// The kernels are specialised at compile time on DBG = "debug plane on or objects hidden";
// the common case (DBG = false) carries no trace of them.  In the DBG = true build the two
// wave-uniform switches are deliberately held in VGPRs (debug_flags), so every branch on them
// is an ordinary exec-masked branch.  Reason (found on gfx950 / ROCm 7.2, see DESIGN.md
// "Compiler hazard"): hipcc may materialise a wave-uniform bool through VALU under a partial
// exec mask inside the divergent march loop and test it again after the loop under a wider
// mask, where the lanes that left the loop early read stale zero bits.
struct DebugFlags
{
	int plane_on; // any(debug normal)
	int show_on;  // any(show_objects)
};
SDF_HD DebugFlags debug_flags(const FrameU &U)
{
	DebugFlags f;
	f.plane_on = U.debug_plane_on;
	f.show_on = U.show_on;
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("" : "+v"(f.plane_on), "+v"(f.show_on));
#endif
	return f;
}
SDF_HD bool frame_needs_debug(const FrameU &U) { return U.debug_plane_on != 0 || U.show_on == 0; }

// ---- scene distance with the driver's debug plane (pshader_sdf.hlsl:111-135) ---------------
template <class Scene, bool DBG>
SDF_HD float map_geometry(const FrameU &U, const DebugFlags &F, const typename Scene::RayInv &R, vec3 p, vec3 dir, bool fast)
{
	if (!DBG)
		return Scene::dist(U, R, p, dir, fast);
	float d = 3e38f;
	if (F.show_on)
		d = Scene::dist(U, R, p, dir, fast);
	if (F.plane_on)
	{
		float plane = sd_plane_fast(p - V3(U.debug_x, U.debug_y, U.debug_z), dir, fast, U.debug_normal);
		return min1(d, plane);
	}
	return d;
}

// ---- the march state a scene's geometry step may look at (pshader_sdf.hlsl:187-218, 297-302) -----------------------
// In the reference a geometry step sees the whole GeometryInput: the running camera_distance of the sample and the pixel's
// ray offsets besides position and direction.  No scene of the reference reads them there, and the scenes compiled ahead of
// time take (p, dir) only.  A scene that declares `static constexpr bool geometry_reads_march_state = true` (what scenes in
// the reference's dialect get, sdfr_hlsl.h) has `dist` called with a sixth argument instead; for every other scene none of
// this exists in the generated code.
struct GeoStep
{
	float camera_distance;       // geometry.camera_distance at this sample (the hit's, for the normal's samples)
	vec3 right_off, bottom_off;  // geometry.right_ray_offset / bottom_ray_offset: the pixel's, for every ray of the pixel
};
template <bool B>
struct VoidIf {};
template <>
struct VoidIf<true> { typedef void type; };
template <class Scene, class = void>
struct SceneReadsMarchState { static constexpr bool value = false; };
template <class Scene>
struct SceneReadsMarchState<Scene, typename VoidIf<Scene::geometry_reads_march_state>::type> { static constexpr bool value = true; };

template <class Scene, bool DBG>
SDF_HD float map_geometry_at(const FrameU &U, const DebugFlags &F, const typename Scene::RayInv &R, vec3 p, vec3 dir, bool fast, const GeoStep &gs)
{
	if (!DBG)
		return Scene::dist(U, R, p, dir, fast, gs);
	float d = 3e38f;
	if (F.show_on)
		d = Scene::dist(U, R, p, dir, fast, gs);
	if (F.plane_on)
	{
		float plane = sd_plane_fast(p - V3(U.debug_x, U.debug_y, U.debug_z), dir, fast, U.debug_normal);
		return min1(d, plane);
	}
	return d;
}

// ---- sphere tracing with over-relaxation (pshader_sdf.hlsl:179-220) as a resumable state ----
struct March
{
	vec3 start, dir;
	float t;          // camera_distance
	float last_d;     // last_scene_distance
	float last_safe;  // last_safe_camera_distance
	float factor;     // step_factor
	float d;          // scene_distance of the last evaluation (already * inside_sign)
	uint32_t iter;
};
enum { MARCH_CONTINUE = 0, MARCH_HIT = 1, MARCH_MISS = 2 };

SDF_HD March march_begin(vec3 start, vec3 dir)
{
	March m;
	m.start = start;
	m.dir = dir;
	m.t = 0.f;
	m.last_d = 0.f;
	m.last_safe = 0.f;
	m.factor = 1.f;
	m.d = 0.f;
	m.iter = 0;
	return m;
}
SDF_HD vec3 march_pos(const March &m) { return mad(m.dir, m.t, m.start); }

// consume one scene-distance sample `d` taken at march_pos(m); requires m.iter < iter_count
SDF_HD int march_advance(March &m, float d, float dist_max, uint32_t iter_count, float dist_eps)
{
	// written with selects instead of branches: the loop body is short and every lane of the wave
	// takes one of the three ways each iteration
	m.d = d;
	// over-stepped: rewind to the last safe point and continue without relaxation
	const bool over = (m.factor > 1.f) & ((m.last_d + d) < m.last_d * m.factor);
	const bool out_of_range = m.t > dist_max;
	const bool stop = !over & (out_of_range | (d < dist_eps));
	const float t_fwd = m.t + d * m.factor;
	const float safe_fwd = m.t + d;
	m.last_d = over ? m.last_d : d;
	m.t = stop ? m.t : (over ? m.last_safe : t_fwd);
	m.last_safe = (over | stop) ? m.last_safe : safe_fwd;
	m.factor = over ? 1.f : m.factor;
	if (stop) return out_of_range ? MARCH_MISS : MARCH_HIT;
	m.iter++;
	return m.iter < iter_count ? MARCH_CONTINUE : MARCH_MISS;
}
// relaxation starts at the fourth sample (pshader_sdf.hlsl:189-192)
SDF_HD void march_pre(March &m) { if (m.iter == 3) m.factor = 1.5f; }

// ---- forward-difference normal (pshader_sdf.hlsl:164-177, Q10) --------------------------------
SDF_HD vec3 grad_sample_pos(vec3 p, int axis, float eps)
{
	if (axis == 0) return p + V3(eps, 0.f, 0.f);
	if (axis == 1) return p + V3(0.f, eps, 0.f);
	return p + V3(0.f, 0.f, eps);
}

// ---- the scene's map_normal (pshader_sdf.hlsl:318-330; sdf_structs.hlsl:39-52) -----------------------
// A scene may declare `static SDF_HD void normal(const FrameU &U, const SurfacePoint &sp, NormalOut &no)`: called once per
// hit with the NormalOutput preloaded {grad_eps, 0, false}; sp.normal is 0 (the geometric normal is what is being made).
// It may hand back the normal itself (no.use_normal = true: the three forward-difference evaluations are not made) and /
// or change no.sample_dist, the spacing of those samples -- which the driver uses a second time, for the offset of the
// shadow rays (pshader_sdf.hlsl:520).  Every scene of the reference leaves map_normal empty: a scene without the
// member costs nothing.
template <class...>
struct VoidOfN { typedef void type; };
template <class Scene, class = void>
struct SceneNormal
{
	static constexpr bool available = false;
	static SDF_HD void call(const FrameU &, const SurfacePoint &, NormalOut &) {}
};
template <class Scene>
struct SceneNormal<Scene, typename VoidOfN<decltype(&Scene::normal)>::type>
{
	static constexpr bool available = true;
	static SDF_HD void call(const FrameU &U, const SurfacePoint &sp, NormalOut &no) { Scene::normal(U, sp, no); }
};
// normal_output as the driver preloads it, then the scene's say
template <class Scene>
SDF_HD NormalOut scene_normal(const FrameU &U, vec3 hit_pos, vec3 dir, float camera_distance, vec3 right_ray, vec3 bottom_ray)
{
	NormalOut no;
	no.sample_dist = U.grad_eps;
	no.normal = V3s(0.f);
	no.use_normal = false;
	if (SceneNormal<Scene>::available)
	{
		SurfacePoint sp;
		sp.pos = hit_pos;
		sp.dir = dir;
		sp.camera_distance = camera_distance;
		sp.right_off = right_ray;
		sp.bottom_off = bottom_ray;
		sp.normal = V3s(0.f);
		sp.iteration_count = 0u;
		sp.scene_distance = 0.f;
		SceneNormal<Scene>::call(U, sp, no);
	}
	return no;
}

// ---- the scene's map_light as ONE call per hit (pshader_sdf.hlsl:505-516) -----------------------------
// The scenes compiled ahead of time answer `light(U, i, L)` slot by slot and `ambient()` without arguments: their lights do
// not depend on the hit point, and unused slots cost nothing.  A scene may instead declare
//   static SDF_HD void lights(const FrameU &U, const SurfacePoint &sp, Light L[SDFR_MAX_LIGHTS], bool used[SDFR_MAX_LIGHTS], float &ambient)
// -- the reference's own shape: map_light(geometry, inout LightOutput[8], inout ambient), called once per lit hit with
// `used` all false and `ambient` 0.075 (what scenes in the reference's dialect get, sdfr_hlsl.h).
template <class Scene, class = void>
struct SceneLights
{
	static constexpr bool available = false;
	static SDF_HD void call(const FrameU &, const SurfacePoint &, Light *, bool *, float &) {}
};
template <class Scene>
struct SceneLights<Scene, typename VoidOfN<decltype(&Scene::lights)>::type>
{
	static constexpr bool available = true;
	static SDF_HD void call(const FrameU &U, const SurfacePoint &sp, Light *L, bool *used, float &ambient) { Scene::lights(U, sp, L, used, ambient); }
};
template <class Scene, bool HasLights = SceneLights<Scene>::available>
struct SceneAmbient { static SDF_HD float get() { return Scene::ambient(); } };
template <class Scene>
struct SceneAmbient<Scene, true> { static SDF_HD float get() { return 0.075f; } };
template <class Scene, bool HasLights = SceneLights<Scene>::available>
struct SceneLightSlot { static SDF_HD bool get(const FrameU &U, int i, Light &L) { return Scene::light(U, i, L); } };
template <class Scene>
struct SceneLightSlot<Scene, true> { static SDF_HD bool get(const FrameU &, int, Light &) { return false; } };

// ---- results handed from marching to shading ---------------------------------------------------
struct HitInfo
{
	vec3 pos;         // geometry_input.pos at the hit
	float t;          // camera_distance
	float d;          // scene_distance (* inside_sign)
	uint32_t iter;
	vec3 normal;      // normal_output.normal
	float sample_dist; // normal_output.normal_sample_dist: grad_eps unless the scene's map_normal changed it (pshader_sdf.hlsl:323,520)
};

// the queue storage is supplied by the caller (registers/scratch in the per-pixel kernel,
// HBM arrays in the wavefront kernels)
template <class Store>
struct Spawner
{
	Store &store;
	uint64_t depths;
	int count;
	int slots;
	SDF_HD Spawner(Store &s, uint64_t d, int c, int n) : store(s), depths(d), count(c), slots(n) {}
	// `if (ray_count < RAY_COUNT) { find_free_ray; fill; ++ray_count; }`
	SDF_HD void push(const RayRec &r)
	{
		if (count < slots)
		{
			int slot = queue_free(depths, slots);
			store.put(slot, r);
			depths = queue_set_depth(depths, slot, ray_depth(r));
			++count;
		}
	}
};

SDF_HD Material default_material(const FrameU &U, vec3 hit_pos)
{
	Material m;
	m.id = MAT_NONE;
	m.mpos = hit_pos;
	m.prop_x = 0.f;
	m.diffuse = V4(0.f, 0.f, 0.f, 1.f);
	m.specular = V4(0.f, 0.f, 0.f, 60.f);
	m.emissive = V3s(0.f);
	m.reflection = V3s(0.f);
	m.refraction = V3s(0.f);
	m.ior = 1.4f;
	m.normal = V4(0.f, 0.f, 0.f, 0.f);
	m.max_cost = U.max_cost_default;
	m.use_hdr = true;
	return m;
}

// map_material (pshader_sdf.hlsl:137-162)
template <class Scene, bool DBG>
SDF_HD void map_material(const FrameU &U, const DebugFlags &F, const SurfacePoint &sp, Material &m)
{
	if (!DBG)
	{
		Scene::material(U, sp, m);
		return;
	}
	bool on_debug_plane = false;
	if (F.plane_on)
	{
		float plane = sd_plane(sp.pos - V3(U.debug_x, U.debug_y, U.debug_z), U.debug_normal);
		on_debug_plane = on_surface(U, plane);
	}
	if (on_debug_plane)
	{
		RayFlags nf;
		nf.has_transparent = false;
		nf.is_shadow = false;
		nf.last_transparent_pos = V3s(0.f);
		typename Scene::RayInv R0 = Scene::ray_setup(U, sp.dir, nf);
		float d;
		if constexpr (SceneReadsMarchState<Scene>::value)
		{
			GeoStep gs;
			gs.camera_distance = sp.camera_distance;
			gs.right_off = sp.right_off;
			gs.bottom_off = sp.bottom_off;
			d = Scene::dist(U, R0, sp.pos, sp.dir, false, gs);
		}
		else
			d = Scene::dist(U, R0, sp.pos, sp.dir, false);
		m.id = MAT_DISTANCE_PLANE;
		m.prop_x = d / U.debug_scale;
	}
	else
	{
		Scene::material(U, sp, m);
	}
}

// A scene may declare `static SDF_HD bool ray_escapes(const FrameU &U, const RayInv &R, vec3 p, vec3 dir)`: true only if
// NOTHING of the scene lies on the ray from p onwards -- then the ray is a miss already (an escaped shadow ray delivers
// its light, any other ray sees the background, whose colour does not depend on the step count in such a scene), and the
// pixel kernel stops marching it (FrameU::step_shortcuts; never in the debug-plane build, whose plane is an extra object).
template <class Scene, class = void>
struct RayEscapes
{
	static constexpr bool available = false;
	template <class R> static SDF_HD bool test(const FrameU &, const R &, vec3, vec3) { return false; }
};
template <class Scene>
struct RayEscapes<Scene, typename VoidOfN<decltype(&Scene::ray_escapes)>::type>
{
	static constexpr bool available = true;
	template <class R> static SDF_HD bool test(const FrameU &U, const R &r, vec3 p, vec3 dir) { return Scene::ray_escapes(U, r, p, dir); }
};

// ... and / or `static SDF_HD float escapes_from(const FrameU &U, vec3 start, vec3 dir, float range)`, worked out once per ray: the
// distance along the ray from which on nothing of the scene lies between the march and `range`, where it declares the ray out of
// range (a shadow ray towards a point light ends there; samples beyond are misses whatever they measure) -- 3e38 if there is no such
// distance.  A sample beyond it is a miss as well; one comparison per step.
template <class Scene, class = void>
struct EscapesFrom
{
	static constexpr bool available = false;
	static SDF_HD float get(const FrameU &, vec3, vec3, float) { return 3e38f; }
};
template <class Scene>
struct EscapesFrom<Scene, typename VoidOfN<decltype(&Scene::escapes_from)>::type>
{
	static constexpr bool available = true;
	static SDF_HD float get(const FrameU &U, vec3 start, vec3 dir, float range) { return Scene::escapes_from(U, start, dir, range); }
};
// A scene with ray_escapes may also declare `static constexpr bool inline_escaped_shadows = true`: a shadow ray that escapes where it
// starts (no evaluation at all: two thirds of the rays of gems with eight lights, whose floor pixels send eight of them past the
// ring) is then not queued -- a 48-byte record out and back, a turn of the bounce loop -- but delivers its light from the light
// loop that made it, when that keeps the order of the pixel's sums: the queue is empty (it would be the next ray taken: smallest
// depth, lowest slot), the hit's own colour has been added (shade_hit adds it before the lights' rays: InlineShadows::acc), and
// the ray budget reaches it.  From the first ray that has to be queued on, the rest are queued as well.
template <class Scene, class = void>
struct InlineEscapedShadows { static constexpr bool value = false; };
template <class Scene>
struct InlineEscapedShadows<Scene, typename VoidOfN<decltype(Scene::inline_escaped_shadows)>::type>
{
	static constexpr bool value = Scene::inline_escaped_shadows && RayEscapes<Scene>::available;
};
struct InlineShadows
{
	vec3 acc;   // the pixel's sum; shade_hit adds the hit's colour and the delivered light to it, in the order the bounce loop would
	int budget; // turns of the bounce loop left after this ray's
	int taken;  // shadow rays delivered here: each counts as a ray and a turn
};

// light i of the frame: the scene's (its own slot function or the table a scene in the reference's shape filled), or an extension light
template <class Scene, class Table, class Used>
SDF_HD bool light_in_slot(const FrameU &U, int i, const Table &table, const Used &table_used, Light &L)
{
	bool used = SceneLightSlot<Scene>::get(U, i, L);
	if (SceneLights<Scene>::available)
	{
		L = table[i];
		used = table_used[i];
	}
	if (i >= 1 && i <= U.extension_lights) // extension: orbiting point lights (sdfr_frame.h)
	{
		const float *E = U.ext_light[i - 1];
		L.pos = V3(E[0], E[1], E[2]);
		L.directional = false;
		L.extend = 0.25f;
		L.falloff = 0.25f;
		L.color = V3(E[3], E[4], E[5]);
		used = true;
	}
	return used;
}
SDF_HD vec3 light_colour(const Light &L)
{
	float falloff = 1.f;
	if (!L.directional) falloff = pow1(0.1f, L.falloff); // distance-independent (Q3)
	return L.color * falloff;
}

// Shade a ray that hit the scene (pshader_sdf.hlsl:317-620).  Returns the colour this ray
// adds to the pixel (already multiplied by the ray's contribution); updates `hdr`; pushes
// secondary rays.
template <class Scene, bool DBG, class Store, bool INL = false>
SDF_HD vec3 shade_hit(const FrameU &U, const DebugFlags &F, const RayRec &ray, const PixelRay &px, const HitInfo &hit, float max_range,
	float &hdr, Spawner<Store> &q, InlineShadows *inl = nullptr)
{
	bool added = false; // INL: the returned colour is in inl->acc already
	const uint32_t depth = ray_depth(ray);
	const float inside_sign = ray_inside_sign(ray);
	const vec3 view_dir = ray.dir;

	SurfacePoint sp;
	sp.pos = hit.pos;
	sp.dir = ray.dir;
	sp.camera_distance = hit.t;
	sp.right_off = px.right_ray;
	sp.bottom_off = px.bottom_ray;
	sp.normal = hit.normal;
	sp.iteration_count = hit.iter;
	sp.scene_distance = hit.d;

	Material m = default_material(U, hit.pos);
	map_material<Scene, DBG>(U, F, sp, m);
	// extension, off by default: reflective marble (sdfr_limits::extension_marble_reflection)
	if (U.extension_marble_reflection != 0.f && (m.id == MAT_MARBLE_DARK || m.id == MAT_MARBLE_LIGHT)) m.reflection = V3s(U.extension_marble_reflection);

	vec3 out = V3s(0.f);
	if (!ray_is_shadow(ray))
	{
		// the first surface decides whether the pixel is tone-mapped (Q2)
		float new_hdr = m.use_hdr ? 1.f : 0.f;
		hdr = lerp1(hdr, new_hdr, step1(hdr, 0.f));

		vec3 n = lerp(hit.normal, V3(m.normal.x, m.normal.y, m.normal.z), m.normal.w);

		// reflection: cost 3, outside rays only
		if (any3(m.reflection) && inside_sign > 0.f && depth + 3 < m.max_cost)
		{
			vec3 rv = reflect(view_dir, n);
			RayRec c;
			c.pos = mad(rv, U.reflect_eps, hit.pos);
			c.dir = rv;
			c.contrib = m.reflection * ray.contrib;
			c.shadow_range = 0.f;
			c.bits = depth + 3;
			q.push(c);
		}
		// refraction: admitted at cost 4, queued at cost 2 (pshader_sdf.hlsl:387,405)
		if (any3(m.refraction) && depth + 4 < m.max_cost)
		{
			RayRec c;
			vec3 rv;
			if (inside_sign > 0.f)
			{
				rv = refract(view_dir, n, 1.f / m.ior);
				c.bits = (depth + 2) | RAY_INSIDE;
			}
			else
			{
				rv = refract(view_dir, -n, m.ior);
				c.bits = depth + 2;
			}
			c.pos = mad(rv, U.refract_eps, hit.pos);
			c.dir = rv;
			c.contrib = m.refraction * ray.contrib;
			c.shadow_range = 0.f;
			q.push(c);
		}

		vec3 diffuse = V3(m.diffuse.x, m.diffuse.y, m.diffuse.z);
		vec3 color = V3s(0.f);
		bool use_light = true;
		switch (m.id)
		{
		case MAT_ITER:
			color = color + mat_iter_heat(hit.iter, (uint32_t)(U.iter_count - 1));
			use_light = false;
			hdr = 0.f;
			break;
		case MAT_PLAIN:
			color = color + diffuse;
			use_light = false;
			break;
		case MAT_NORMAL1:
		{
			vec3 nc = max(n, 0.01f);
			nc = nc / max1(max1(nc.x, nc.y), nc.z);
			color = color + nc;
			use_light = false;
			hdr = 0.f;
			break;
		}
		case MAT_NORMAL2:
			color = color + abs(n);
			use_light = false;
			hdr = 0.f;
			break;
		case MAT_DISTANCE_PLANE:
			color = color + mat_debug_plane(m.prop_x);
			use_light = false;
			hdr = 0.f;
			break;
		case MAT_WOOD:
			diffuse = diffuse + mat_wood(m.mpos);
			break;
		case MAT_MARBLE_DARK:
			diffuse = diffuse + mat_marble(m.mpos, V3(0.556f, 0.478f, 0.541f));
			break;
		case MAT_MARBLE_LIGHT:
			diffuse = diffuse + mat_marble(m.mpos, V3(0.7f, 0.7f, 0.7f));
			break;
		case MAT_FIRE:
		{
			float fadeout = sat1(dot(-view_dir, n));
			vec4 fc = mat_fire(m.mpos, 1.f - fadeout);
			color = color + V3(fc.x, fc.y, fc.z);
			m.diffuse = V4(1.f, 1.f, 1.f, sat1(fc.w));
			break;
		}
		default:
			break;
		}

		// see-through surface: continue the same ray from the hit point (Q8)
		if (m.diffuse.w < 1.f && depth + 2 < m.max_cost)
		{
			RayRec c;
			c.pos = hit.pos;
			c.dir = view_dir;
			c.contrib = (1.f - m.diffuse.w) * V3(m.diffuse.x, m.diffuse.y, m.diffuse.z) * ray.contrib;
			c.shadow_range = 0.f;
			c.bits = (depth + 2) | RAY_TRANSPARENT;
			q.push(c);
		}

		if (use_light)
		{
			float ambient = SceneAmbient<Scene>::get(); // map_light may override the default 0.075
			// a scene in the reference's own shape fills the whole light table in one call (SceneLights)
			Light table[SceneLights<Scene>::available ? SDFR_MAX_LIGHTS : 1];
			bool table_used[SceneLights<Scene>::available ? SDFR_MAX_LIGHTS : 1];
			if (SceneLights<Scene>::available)
			{
				for (int i = 0; i < SDFR_MAX_LIGHTS; ++i)
				{
					table_used[i] = false;
					table[i].pos = V3s(0.f);
					table[i].directional = false;
					table[i].extend = 0.f;
					table[i].color = V3s(0.f);
					table[i].falloff = 0.f;
				}
				SceneLights<Scene>::call(U, sp, table, table_used, ambient);
			}
			float move = max1(U.shadow_eps, hit.sample_dist) + max1(0.f, -hit.d);
			vec3 lit_pos = mad(n, move, hit.pos);
			const float alpha = sat1(m.diffuse.w);

			if constexpr (INL)
			{
				// The hit's own colour first -- every light's ambient share, in the lights' order -- so that it is in the pixel's sum
				// before any light a shadow ray delivers; then the loop as below, the ambient line left out.
				for (int i = 0; i < U.light_count; ++i)
				{
					Light L;
					if (!light_in_slot<Scene>(U, i, table, table_used, L)) continue;
					color = color + diffuse * light_colour(L) * ambient;
				}
				color = color + m.emissive;
				color = color * alpha;
				out = out + color * ray.contrib;
				inl->acc = inl->acc + out;
				added = true;
				for (int i = 0; i < U.light_count; ++i)
				{
					Light L;
					if (!light_in_slot<Scene>(U, i, table, table_used, L)) continue;
					vec3 ldir;
					float trace_dist;
					if (L.directional)
					{
						ldir = L.pos / (length(L.pos) + U.dist_eps);
						trace_dist = U.range;
					}
					else
					{
						ldir = lit_pos - L.pos;
						trace_dist = length(ldir);
						ldir = ldir / trace_dist;
						trace_dist = trace_dist - L.extend;
					}
					const vec3 lcol = light_colour(L);
					float ndl = sat1(dot(-n, ldir));
					vec3 direct = V3s(0.f) + diffuse * lcol * ndl;
					vec3 half_vec = -normalize(view_dir + ldir);
					float spec = pow1(sat1(dot(n, half_vec)), m.specular.w);
					direct = direct + V3(m.specular.x, m.specular.y, m.specular.z) * lcol * spec;
					if (depth + 2 < m.max_cost && ndl > 0.f)
					{
						RayRec c;
						c.pos = lit_pos;
						c.dir = -ldir;
						c.contrib = direct * ray.contrib * alpha;
						c.shadow_range = trace_dist;
						c.bits = (depth + 2) | RAY_SHADOW;
						// the reference's queue holds the rays delivered so far as well: it has a slot for this one only if all of them fit
						if (inl->taken + q.count >= q.slots) continue;
						if (U.step_shortcuts != 0 && q.count == 0 && inl->taken < inl->budget)
						{
							// the first turn of the march loop (render_pixel): an unrelaxed sample at the ray's start
							const typename Scene::RayInv R = Scene::ray_setup(U, c.dir, ray_flags(c));
							March sm = march_begin(c.pos, c.dir);
							march_pre(sm);
							if (sm.factor == 1.f && RayEscapes<Scene>::test(U, R, march_pos(sm), c.dir))
							{
								inl->acc = inl->acc + (V3s(0.f) + c.contrib); // shade_miss of a shadow ray
								inl->taken++;
								q.store.ray_marched(c, 0u, MARCH_MISS);
								continue;
							}
						}
						q.push(c);
					}
				}
			}
			else // the reference's one loop: every light's ambient share and its shadow ray (body not re-indented: the headline kernel's code)
			for (int i = 0; i < U.light_count; ++i)
			{
				Light L;
				bool used = SceneLightSlot<Scene>::get(U, i, L);
				if (SceneLights<Scene>::available)
				{
					L = table[i];
					used = table_used[i];
				}
				if (i >= 1 && i <= U.extension_lights) // extension: orbiting point lights (sdfr_frame.h)
				{
					const float *E = U.ext_light[i - 1];
					L.pos = V3(E[0], E[1], E[2]);
					L.directional = false;
					L.extend = 0.25f;
					L.falloff = 0.25f;
					L.color = V3(E[3], E[4], E[5]);
					used = true;
				}
				if (!used) continue;
				vec3 ldir;
				float trace_dist;
				float falloff = 1.f;
				if (L.directional)
				{
					ldir = L.pos / (length(L.pos) + U.dist_eps);
					trace_dist = U.range;
				}
				else
				{
					ldir = lit_pos - L.pos;
					trace_dist = length(ldir);
					ldir = ldir / trace_dist;
					trace_dist = trace_dist - L.extend;
					falloff = pow1(0.1f, L.falloff); // distance-independent (Q3)
				}
				vec3 lcol = L.color * falloff;

				color = color + diffuse * lcol * ambient;

				// diffuse + Blinn specular are delivered by the shadow ray if it escapes (F6)
				float ndl = sat1(dot(-n, ldir));
				vec3 direct = V3s(0.f) + diffuse * lcol * ndl;
				vec3 half_vec = -normalize(view_dir + ldir);
				float spec = pow1(sat1(dot(n, half_vec)), m.specular.w);
				direct = direct + V3(m.specular.x, m.specular.y, m.specular.z) * lcol * spec;

				if (depth + 2 < m.max_cost && ndl > 0.f)
				{
					RayRec c;
					c.pos = lit_pos;
					c.dir = -ldir;
					c.contrib = direct * ray.contrib * alpha;
					c.shadow_range = trace_dist;
					c.bits = (depth + 2) | RAY_SHADOW;
					q.push(c);
				}
			}
			if (!added)
			{
				color = color + m.emissive;
				color = color * alpha;
			}
		}
		if (!added) out = out + color * ray.contrib;
	}
	else
	{
		// a shadow ray stopped by a see-through surface continues, tinted (pshader_sdf.hlsl:598-619)
		if (m.diffuse.w < 1.f && depth + 2 < m.max_cost)
		{
			RayRec c;
			c.pos = hit.pos;
			c.dir = view_dir;
			c.contrib = (1.f - m.diffuse.w) * V3(m.diffuse.x, m.diffuse.y, m.diffuse.z) * ray.contrib;
			c.shadow_range = max_range - hit.t;
			c.bits = (depth + 2) | RAY_TRANSPARENT | RAY_SHADOW;
			q.push(c);
		}
	}
	if constexpr (INL)
		if (!added) inl->acc = inl->acc + out;
	return out;
}

// A ray that left the scene (pshader_sdf.hlsl:621-632): an escaped shadow ray delivers the
// light it carries, any other ray sees the background.
template <class Scene>
SDF_HD vec3 shade_miss(const FrameU &U, const RayRec &ray, uint32_t iter)
{
	if (ray_is_shadow(ray))
		return V3s(0.f) + ray.contrib;
	return V3s(0.f) + Scene::background(U, ray.dir, iter) * ray.contrib;
}

} // namespace sdfr
