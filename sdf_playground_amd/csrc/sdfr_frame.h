// sdfr_frame.h -- per-frame uniform block and the small records exchanged between the
// pipeline stages (device code, host-compilable).
#pragma once
#include "sdfr_math.h"

namespace sdfr {

enum { SDFR_MAX_RAYS = 8, SDFR_MAX_LIGHTS = 8, SDFR_MAX_SCENE_VARS = 8, SDFR_SCENE_UNIFORMS = 48 };

// Frame uniforms = the reference's camera constant buffer b0 (pshader_sdf.hlsl:17-26), its
// VAR_ buffer b1 (ShaderUtil.cpp:193-267), the driver's compile-time limits
// (pshader_sdf.hlsl:60-64,350) as run-time values, plus values derived once per frame on the
// host.  Passed to kernels by value: it is wave-uniform, so it lives in SGPRs.
struct FrameU
{
	vec3 eye, front, right, top;
	float stime;
	int width, height;
	int iter_count, bounce_count, ray_count, light_count;
	float range;
	uint32_t max_cost_default;
	// driver variables (pshader_sdf.hlsl:88-108,142)
	float debug_nx, debug_ny, debug_nz, debug_scale, debug_x, debug_y, debug_z, show_objects;
	float scene_var[SDFR_MAX_SCENE_VARS];
	// ---- derived on the host (sdfr_derive_frame) ----
	vec3 debug_normal;   // normalised debug-plane normal, or 0
	int debug_plane_on;  // any(debug normal)
	int show_on;         // any(show_objects)
	float ddx, ddy;      // screen-space derivatives of the NDC coordinate: 2/W, -2/H
	float sky_s, sky_c;  // sin/cos(-stime * 0.025) for the shared sky
	float su[SDFR_SCENE_UNIFORMS]; // scene-specific constants (Scene::prepare)
	// EXTENSION, not in the reference (SURVEY.md 8d cfg 5, "8 lights"): n orbiting point lights
	// that overwrite slots 1..n of the scene's light table; position xyz + colour rgb, derived on
	// the host per frame (sdfr_hostframe.h).  0 = reference behaviour.
	int extension_lights;
	float ext_light[SDFR_MAX_LIGHTS - 1][6];
	// EXTENSION, not in the reference (SURVEY.md 8d cfg 3 "2 reflection bounces"): reflection colour given to
	// marble materials; 0 = reference behaviour
	float extension_marble_reflection;
	// Not a change of any pixel: a ray for which the scene says that nothing lies ahead any more (Scene::ray_escapes)
	// is booked as the miss it is going to be without marching the remaining steps.  The per-pixel count of march
	// evaluations then falls short of the reference's; 0 = march every step (the counters equal the oracle's).
	int step_shortcuts;
	// the driver's five epsilons (pshader_sdf.hlsl:31-35: dist_eps 1e-4 "how close to the object before terminating" -- the
	// hit test, the MATERIAL / OBJECT_TRANSPARENT macros, sdSphereFast, the directional light's normalisation --, grad_eps
	// 1e-4, reflect_eps 1e-3, refract_eps 1e-3, shadow_eps 3e-4) as run-time values (sdfr_limits); defaults = reference
	float dist_eps, grad_eps, reflect_eps, refract_eps, shadow_eps;
	// (float)width, (float)height (frame_derive): the vector unit is the only one that converts, and a kernel that made them
	// itself would keep two vector registers on them for as long as a persistent wave lives
	float widthf, heightf;
};

// One queued ray, 11 dwords.  last_transparent_pos of the reference's Ray struct
// (pshader_sdf.hlsl:40-51) is not stored: it is `pos` when has_transparent is set and 0
// otherwise (pshader_sdf.hlsl:377,401,415,494,578,610).
struct RayRec
{
	vec3 pos, dir, contrib;
	float shadow_range;
	uint32_t bits; // [7:0] depth cost, [8] inside (inside_sign = -1), [9] has_transparent, [10] is_shadow_ray
};
enum { RAY_INSIDE = 1u << 8, RAY_TRANSPARENT = 1u << 9, RAY_SHADOW = 1u << 10, RAY_DEPTH_MASK = 0xffu, RAY_DEPTH_INVALID = 0xffu };

SDF_HD uint32_t ray_depth(const RayRec &r) { return r.bits & RAY_DEPTH_MASK; }
SDF_HD float ray_inside_sign(const RayRec &r) { return (r.bits & RAY_INSIDE) ? -1.f : 1.f; }
SDF_HD bool ray_is_shadow(const RayRec &r) { return (r.bits & RAY_SHADOW) != 0; }
SDF_HD bool ray_has_transparent(const RayRec &r) { return (r.bits & RAY_TRANSPARENT) != 0; }

// the MarchingInput of the scene ABI (sdf_structs.hlsl:23-37) as derived from a queued ray
struct RayFlags { bool has_transparent; bool is_shadow; vec3 last_transparent_pos; };

// what a scene's material callback sees of the hit point (GeometryInput with dir.w = 0)
struct SurfacePoint
{
	vec3 pos, dir;
	float camera_distance;
	vec3 right_off, bottom_off; // pixel footprint per unit of distance
	vec3 normal;                // MaterialInput.obj_normal: the geometric normal at the hit
	uint32_t iteration_count;   // MaterialInput.iteration_count: march iterations of the ray that hit (sdf_structs.hlsl:54-64)
	float scene_distance;       // MaterialInput.scene_distance: the distance the march stopped at (already * inside_sign)
};

// NormalOutput of the scene ABI (sdf_structs.hlsl:39-52), preloaded by the driver (pshader_sdf.hlsl:320-323):
// use_normal false, normal 0, sample_dist = grad_eps
struct NormalOut
{
	float sample_dist; // normal_sample_dist: spacing of the forward-difference samples; "larger than usual values lead to rounded corners"
	vec3 normal;       // the scene's own normal
	bool use_normal;   // true: take `normal` instead of sampling
};

// MaterialOutput of the scene ABI (sdf_structs.hlsl:66-110), defaults of pshader_sdf.hlsl:338-351
struct Material
{
	uint32_t id;
	vec3 mpos;       // material_position.xyz
	float prop_x;    // material_properties.x
	vec4 diffuse;    // rgb + alpha
	vec4 specular;   // rgb + power
	vec3 emissive, reflection, refraction;
	float ior;
	vec4 normal;     // xyz + blend
	uint32_t max_cost;
	bool use_hdr;
};

// LightOutput of the scene ABI (sdf_structs.hlsl:112-130)
struct Light
{
	vec3 pos;
	bool directional; // pos.w == 1
	float extend;
	vec3 color;
	float falloff;
};

// device-side totals of one render
struct RenderTotals
{
	unsigned long long pixels, rays, march_evals, hits;
};

// Which rows of the full frame this launch renders.  The frame is cut into 8-row strips.  Of
// every `priv_period` consecutive strips the first `priv_count` are PRIVATE to rank 0 -- it renders
// them straight into the final image (they never travel) -- and the others are SHARED: dealt
// round-robin over all ranks into compact buffers that are gathered.  priv_count = 0 (the default)
// is plain round-robin: local row l is global row ((l / 8) * world + rank) * 8 + l % 8; world = 1,
// rank = 0 is then the whole frame.
struct RowMap
{
	int local_rows;
	int rank, world;
	int tile_w_log2; // a wave covers a (1 << tile_w_log2) x (64 >> tile_w_log2) pixel tile; 3..6
	int priv_count, priv_period; // 0 <= priv_count < priv_period
	int direct;                  // 1: this launch renders the private strips, pixel index = position in the full image
	// tiles per tile row of the launch, and floor(2^32 / tiles_x): a wave splits its tile index into row and column with
	// scalar multiplies (tile_row_and_column) instead of the vector unit's integer division; set by the launcher (row_map_tiles)
	uint32_t tiles_x, tiles_x_magic;
	// Hand-out units of a persistent full-frame launch (row_map_units): the frame's tiles in squares of (1 << unit_log2)^2 tiles, handed out
	// unit by unit (dearest first by last frame's cost: the feedback of sdfr_pixel_kernel.h is kept per UNIT, not per tile row -- an object in
	// the middle of the picture is dear in the middle of many rows).  unit_log2 = 0: no units, tile rows as they are.
	uint32_t unit_log2, units_x, units_x_magic, units;
	// persistent launches: a wave ends after this many tiles and a fresh one takes its place (0 = never); set by the launcher
	uint32_t retire_after;
	// row feedback (sdfr_pixel_kernel.h): what the launch's row order is valid for -- scene, frame size, which rows this launch
	// renders (rank / world / strip split / private strips), tile shape -- folded into one word by the launcher (0 = no feedback);
	// an order made by a launch with another key is not used
	uint32_t feedback_key;
};

SDF_HD void row_map_tiles(RowMap &rm, int width)
{
	rm.tiles_x = ((uint32_t)width + (1u << rm.tile_w_log2) - 1u) >> rm.tile_w_log2;
	rm.tiles_x_magic = rm.tiles_x > 1u ? (uint32_t)(0x100000000ull / rm.tiles_x) : 0xffffffffu;
}
// squares of tiles such that the frame has at most `max_units` of them; not for strip launches (their rows are not the frame's)
SDF_HD void row_map_units(RowMap &rm, uint32_t max_units)
{
	rm.unit_log2 = 0u;
	rm.units_x = rm.units_x_magic = rm.units = 0u;
	if (rm.world != 1 || rm.priv_count != 0 || rm.direct != 0) return;
	const uint32_t th_log2 = 6u - (uint32_t)rm.tile_w_log2;
	const uint32_t tiles_y = ((uint32_t)rm.local_rows + (1u << th_log2) - 1u) >> th_log2;
	for (uint32_t l = 1u; l <= 8u; ++l)
	{
		const uint32_t ux = (rm.tiles_x + (1u << l) - 1u) >> l, uy = (tiles_y + (1u << l) - 1u) >> l;
		if (ux * uy <= max_units)
		{
			rm.unit_log2 = l;
			rm.units_x = ux;
			rm.units_x_magic = ux > 1u ? (uint32_t)(0x100000000ull / ux) : 0xffffffffu;
			rm.units = ux * uy;
			return;
		}
	}
}
// n / d and n % d with m = floor(2^32 / d): the estimate mulhi(n, m) is the quotient or one below it
SDF_HD void split_by_magic(uint32_t n, uint32_t d, uint32_t magic, uint32_t &quotient, uint32_t &rest)
{
	uint32_t q = (uint32_t)(((unsigned long long)n * magic) >> 32);
	uint32_t r = n - q * d;
	if (r >= d)
	{
		q += 1u;
		r -= d;
	}
	quotient = q;
	rest = r;
}
// tile / tiles_x and tile % tiles_x: with m = floor(2^32 / d) the estimate mulhi(n, m) is the quotient or one below it
SDF_HD void tile_row_and_column(const RowMap &rm, uint32_t tile, uint32_t &row, uint32_t &column)
{
	uint32_t q = (uint32_t)(((unsigned long long)tile * rm.tiles_x_magic) >> 32);
	uint32_t r = tile - q * rm.tiles_x;
	if (r >= rm.tiles_x)
	{
		q += 1u;
		r -= rm.tiles_x;
	}
	row = q;
	column = r;
}

// local strip index of this launch -> strip index in the frame
SDF_HD uint32_t strip_local_to_global(const RowMap &rm, uint32_t ls)
{
	if (rm.direct) return (ls / (uint32_t)rm.priv_count) * (uint32_t)rm.priv_period + ls % (uint32_t)rm.priv_count;
	const uint32_t t = ls * (uint32_t)rm.world + (uint32_t)rm.rank; // index among the shared strips
	if (rm.priv_count == 0) return t;
	const uint32_t shared = (uint32_t)(rm.priv_period - rm.priv_count);
	return (t / shared) * (uint32_t)rm.priv_period + (uint32_t)rm.priv_count + t % shared;
}
// how many strips of a frame of `strips` strips are private
SDF_HD uint32_t private_strip_count(uint32_t strips, int priv_count, int priv_period)
{
	if (priv_count <= 0) return 0;
	const uint32_t rem = strips % (uint32_t)priv_period;
	return (strips / (uint32_t)priv_period) * (uint32_t)priv_count + (rem < (uint32_t)priv_count ? rem : (uint32_t)priv_count);
}

// Everything a launch of the pixel kernel is given, as ONE by-value kernel argument: the frame uniforms first.
struct PixelKernelArgs
{
	FrameU U;
	RowMap rm;
	uint32_t n_work;
	int format;
	void *out;
	uint32_t *pixel_stats;
	RenderTotals *partials, *totals;
	float *ray_queue;
	unsigned long cap; // pixels the ray queue is allocated for (size_t)
	uint32_t *tile_cursors;
};

enum { FORMAT_RGBA32F = 0, FORMAT_RGBA16F = 1, FORMAT_STRIP_RGB32F_A8 = 2, FORMAT_STRIP_RGB16F_A8 = 3 };

} // namespace sdfr
