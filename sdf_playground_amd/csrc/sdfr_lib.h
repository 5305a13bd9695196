// sdfr_lib.h -- SDF primitives, domain operators, floor/sky helpers and procedural
// materials used by the scene functors (device code, host-compilable).
//
// Functionality mirrors the reference shader libraries so that scenes keep their meaning:
//   primitives  Engine/shader/sdf_primitives.hlsl:6-131
//   operators   Engine/shader/sdf_ops.hlsl:6-134
//   floor/sky   Engine/shader/sdf_common.hlsl:4-94
//   materials   Engine/shader/sdf_materials.hlsl:6-31,143-201
// Per-frame uniform sines/cosines are passed in pre-computed (see Scene::prepare).
#pragma once
#include "sdfr_math.h"
#include "sdfr_frame.h"
#include "sdfr_noise.h"

namespace sdfr {

// epsilons of the reference driver (pshader_sdf.hlsl:31-35): the defaults of FrameU::dist_eps ... shadow_eps, which is what
// the code reads (run-time values since round 3, sdfr_limits)
#define SDFR_DEFAULT_DIST_EPS 0.0001f
#define SDFR_DEFAULT_GRAD_EPS 0.0001f
#define SDFR_DEFAULT_REFLECT_EPS 0.001f
#define SDFR_DEFAULT_REFRACT_EPS 0.001f
#define SDFR_DEFAULT_SHADOW_EPS 0.0003f
// the largest dist_eps sdfr_set_limits accepts: the scenes' culling bounds and escape rules carry 0.01 of slack
#define SDFR_MAX_DIST_EPS 0.001f
#define SDFR_MAX_WAVES_PER_BLOCK 4 // of any kernel that evaluates scenes (checked where the block sizes are defined): per-wave LDS of scene code

#define SDFR_SQRT_HALF 0.70710678118654752f
#define SDFR_SQRT_TWO 1.41421356237309504f
#define SDFR_PI 3.14159265358979323f
#define SDFR_TAU 6.28318530717958647f

// material ids (pshader_sdf.hlsl:67-76)
enum MaterialId
{
	MAT_NONE = 0, MAT_PLAIN = 1, MAT_ITER = 2, MAT_NORMAL1 = 3, MAT_NORMAL2 = 4, MAT_DISTANCE_PLANE = 5,
	MAT_WOOD = 20, MAT_MARBLE_DARK = 21, MAT_MARBLE_LIGHT = 22, MAT_FIRE = 23
};

// the MATERIAL macro (pshader_sdf.hlsl:81)
SDF_HD bool on_surface(const FrameU &U, float d) { return abs1(d) < U.dist_eps; }

// ---- values a whole wave needs alike, computed once --------------------------------------------
// Scene functions that hash the corners / neighbours of the lattice cell a point lies in do the same n independent
// computations in every lane whenever the lanes stand in the same cell -- and the lanes of a wave are neighbouring
// pixels at similar depths, so mostly they do.  WaveShare lets n lanes compute one value each and pass them round
// through LDS (n <= SDFR_WAVE_SHARE_SLOTS floats per wave): same expression per value, same bits.
//     uint32_t rank;
//     if (WaveShare::agree(key.x, key.y, key.z, n, &rank)) {        // wave-uniform: all active lanes hold these bits, >= n lanes active
//         if (rank < n) WaveShare::slots()[rank] = f(rank);         // the n lowest active lanes
//         WaveShare::publish();
//         ... read WaveShare::slots()[0 .. n) ...
//         WaveShare::release();                                     // before the slots are written again
//     } else { every lane computes all n itself }
// Device code only; the host build of a scene (tests/hostsim) takes the else branch.
#if defined(__HIP_DEVICE_COMPILE__)
#define SDFR_WAVE_SHARE_SLOTS 24
struct WaveShare
{
	static __device__ __forceinline__ float *slots()
	{
		__shared__ float share[SDFR_MAX_WAVES_PER_BLOCK][SDFR_WAVE_SHARE_SLOTS];
		return share[threadIdx.x >> 6];
	}
	static __device__ __forceinline__ bool agree(float a, float b, float c, uint32_t n, uint32_t *rank)
	{
		const unsigned long long active = __ballot(1);
		const int ia = __float_as_int(a), ib = __float_as_int(b), ic = __float_as_int(c);
		const bool differs = ia != __builtin_amdgcn_readfirstlane(ia) || ib != __builtin_amdgcn_readfirstlane(ib) || ic != __builtin_amdgcn_readfirstlane(ic);
		*rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(active >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)active, 0u));
		return __ballot(differs) == 0ull && (uint32_t)__popcll(active) >= n;
	}
	static __device__ __forceinline__ void publish()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	static __device__ __forceinline__ void release()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
};
#endif

// ---- primitives ------------------------------------------------------------------------
SDF_HD float sd_sphere(vec3 p, float r) { return length(p) - r; }

// analytic ray/sphere distance when `fast`, exact SDF otherwise (sdf_primitives.hlsl:11-45)
SDF_HD float sd_sphere_fast(vec3 p, vec3 dir, bool fast, float r, float dist_eps)
{
	if (!fast) return sd_sphere(p, r);
	float b = -dot(p, dir);
	float c = dot(p, p) - r * r;
	float disc = b * b - c;
	if (disc < 0.f) return 1e10f;
	float root = sqrt1(disc);
	float t1 = b - root;
	float t2 = b + root;
	if (t1 < -dist_eps) return t2 > 0.f ? t2 : 1e10f;
	return t1;
}

SDF_HD float sd_box(vec3 p, vec3 half_size)
{
	vec3 q = abs(p) - half_size;
	return length(max(q, 0.f)) + min1(max1(q.x, max1(q.y, q.z)), 0.f);
}

SDF_HD float sd_plane(vec3 p, vec3 n) { return dot(p, n); }

// For Scene::ray_escapes of a scene that is a floor at y = 0 plus objects inside the ball (c, radius) and below `top`:
// a ray that does not descend has the floor behind it, and leaves the objects behind for good once it is above `top`,
// or if its line passes the ball at more than the radius, or if the ball lies behind it.  `radius` and `top` carry the
// caller's slack.  The direction is NOT taken to be a unit vector: a shadow ray towards a directional light has length
// |L| / (|L| + dist_eps) (pshader_sdf.hlsl: the reference's own normalisation), 0.9996 at the largest dist_eps, and over a lever
// of ten units that moves the line's distance from the centre by more than the slack (found by the randomised hunt in round 4).
// "Behind it" needs the ray to be ABOVE the floor: the fast plane of a ray that does not descend is p.y / 1e-20
// (ground_dist), which for p.y <= 0 -- a camera under or on the floor, a child ray pushed below it -- is <= 0: the
// reference books a HIT of the floor at that very sample.  From p.y > 1e-20 on the floor's distance is >= 1 here and
// grows with every step of a ray that does not descend (NaN fails the test as well).
SDF_HD bool ray_leaves_floor_and_ball(vec3 p, vec3 dir, float top, vec3 c, float radius)
{
	if (!(dir.y >= 0.f) || !(p.y > 1e-20f)) return false;
	if (p.y > top) return true;
	const vec3 v = p - c;
	const float b = dot(v, dir), vv = dot(v, v), dd = dot(dir, dir);
	return vv > radius * radius && (b >= 0.f || vv * dd - b * b > radius * radius * dd);
}
// the same test for one ball alone: the ray has it behind or passes it at a distance
SDF_HD bool ray_passes_ball(vec3 p, vec3 dir, vec3 c, float radius)
{
	const vec3 v = p - c;
	const float b = dot(v, dir), vv = dot(v, v), dd = dot(dir, dir);
	return vv > radius * radius && (b >= 0.f || vv * dd - b * b > radius * radius * dd);
}

// distance along the ray when `fast` (sdf_primitives.hlsl:59-70)
SDF_HD float sd_plane_fast(vec3 p, vec3 dir, bool fast, vec3 n)
{
	float d = dot(p, n);
	if (fast) return d / (sat1(dot(dir, -n)) + 1e-20f);
	return d;
}

SDF_HD float sd_torus_xy(vec3 p, float r_big, float r_small)
{
	vec2 q = V2(length(V2(p.x, p.y)) - r_big, p.z);
	return length(q) - r_small;
}

SDF_HD float sd_capped_cylinder(vec3 p, float h, float r)
{
	vec2 d = abs(V2(length(V2(p.x, p.z)), p.y)) - V2(r, h);
	return min1(max1(d.x, d.y), 0.f) + length(max(d, 0.f));
}

SDF_HD float sd_round_cone(vec3 p, vec3 a, vec3 b, float r1, float r2)
{
	vec3 ba = b - a;
	float l2 = dot(ba, ba);
	float rr = r1 - r2;
	float a2 = l2 - rr * rr;
	float il2 = rcp1(l2);

	vec3 pa = p - a;
	float y = dot(pa, ba);
	float z = y - l2;
	vec3 xs = pa * l2 - ba * y;
	float x2 = dot(xs, xs);
	float y2 = y * y * l2;
	float z2 = z * z * l2;

	float k = sign1(rr) * rr * rr * x2;
	if (sign1(z) * a2 * z2 > k) return sqrt1(x2 + z2) * il2 - r2;
	if (sign1(y) * a2 * y2 < k) return sqrt1(x2 + y2) * il2 - r1;
	return (sqrt1(x2 * a2 * il2) + y * rr) * il2 - r1;
}

// guard objects: distance along the ray to the wall of a repetition cell
SDF_HD float sd_limit1(float p, float dir, float lim)
{
	return ((step1(0.f, dir) - 0.5f) * lim - p) / dir;
}
SDF_HD float sd_limit2(vec2 p, vec2 dir, vec2 lim)
{
	vec2 t = ((V2(step1(0.f, dir.x), step1(0.f, dir.y)) - 0.5f) * lim - p) / dir;
	return min1(t.x, t.y);
}
SDF_HD float sd_limit3(vec3 p, vec3 dir, vec3 lim)
{
	vec3 t = ((V3(step1(0.f, dir.x), step1(0.f, dir.y), step1(0.f, dir.z)) - 0.5f) * lim - p) / dir;
	return min1(min1(t.x, t.y), t.z);
}

// ---- operators -------------------------------------------------------------------------
SDF_HD float op_rep_lim(float p, float count, float size)
{
	float rounded = size * (rne1(p / size + count / 2.f) - count / 2.f);
	float limit = count * size * 0.5f;
	return p - clamp1(rounded, -limit, limit);
}
SDF_HD vec2 op_rep_lim(vec2 p, vec2 count, vec2 size) { return V2(op_rep_lim(p.x, count.x, size.x), op_rep_lim(p.y, count.y, size.y)); }
SDF_HD vec3 op_rep_lim(vec3 p, vec3 count, vec3 size) { return V3(op_rep_lim(p.x, count.x, size.x), op_rep_lim(p.y, count.y, size.y), op_rep_lim(p.z, count.z, size.z)); }

SDF_HD float op_rep_inf(float p, float size)
{
	float x = p + size * 0.5f;
	return x - size * floor1(x / size) - size * 0.5f;
}
// cell size is one of the verified constants (div_c): pass rsize = 1.0f / size
SDF_HD float op_rep_inf_c(float p, float size, float rsize)
{
	float x = p + size * 0.5f;
	return x - size * floor1(div_c(x, size, rsize)) - size * 0.5f;
}
SDF_HD vec2 op_rep_inf_c(vec2 p, float size, float rsize) { return V2(op_rep_inf_c(p.x, size, rsize), op_rep_inf_c(p.y, size, rsize)); }
SDF_HD vec2 op_rep_inf(vec2 p, vec2 size) { return V2(op_rep_inf(p.x, size.x), op_rep_inf(p.y, size.y)); }
SDF_HD vec3 op_rep_inf(vec3 p, vec3 size) { return V3(op_rep_inf(p.x, size.x), op_rep_inf(p.y, size.y), op_rep_inf(p.z, size.z)); }

// angular repetition; folds `p` into the first sector, returns the sector index
SDF_HD float op_rep_angle(vec2 *p, float count)
{
	float angle = atan21(p->y, p->x);
	float reduced = div_c(angle * count, SDFR_TAU, 1.0f / SDFR_TAU) + 0.5f; // tau is on the verified-divisor list
	float index = floor1(reduced);
	reduced = reduced - index;
	angle = (reduced - 0.5f) * SDFR_TAU / count;
	vec2 sc = sincos1(angle);
	*p = V2(sc.y, sc.x) * length(*p);
	return index;
}

// rotation with pre-computed sine/cosine
SDF_HD vec2 rot2(vec2 p, float s, float c) { return V2(p.x * c - p.y * s, p.x * s + p.y * c); }
SDF_HD vec2 op_rotate(vec2 p, float angle)
{
	vec2 sc = sincos1(angle);
	return rot2(p, sc.x, sc.y);
}

SDF_HD float op_shell(float d, float inner, float outer)
{
	float avg = (outer + inner) * 0.5f;
	float diff = (outer - inner) * 0.5f;
	return abs1(d - avg) - diff;
}

SDF_HD vec2 op_ab2uv(vec2 v) { return V2(v.x + v.y, v.x - v.y) * SDFR_SQRT_HALF; }
SDF_HD float op_chamfer(float a, float b, float size) { return (a + b - size) * SDFR_SQRT_HALF; }
SDF_HD float op_chamfer_merge(float a, float b, float size) { return min1(min1(a, b), op_chamfer(a, b, size)); }

SDF_HD float op_pipe(float a, float b, float size, float count)
{
	vec2 uv = op_ab2uv(V2(a, b));
	float diag = size * SDFR_SQRT_HALF - uv.y;
	diag = fmod1(diag, SDFR_SQRT_TWO * size / count);
	uv.y = size * SDFR_SQRT_HALF - diag;
	vec2 ab = op_ab2uv(uv);
	float a_offset = (count - 1.f) / count;
	return length(V2(ab.x - a_offset * size, ab.y)) - size / count;
}
SDF_HD float op_pipe_merge(float a, float b, float size, float count) { return min1(min1(a, b), op_pipe(a, b, size, count)); }
// size and count are literals whose period sqrt(2)*size/count is a verified constant (div_c)
SDF_HD float op_pipe_c(float a, float b, float size, float count)
{
	vec2 uv = op_ab2uv(V2(a, b));
	float diag = size * SDFR_SQRT_HALF - uv.y;
	const float period = SDFR_SQRT_TWO * size / count;
	diag = fmod_c(diag, period, 1.0f / period);
	uv.y = size * SDFR_SQRT_HALF - diag;
	vec2 ab = op_ab2uv(uv);
	float a_offset = (count - 1.f) / count;
	return length(V2(ab.x - a_offset * size, ab.y)) - size / count;
}
SDF_HD float op_pipe_merge_c(float a, float b, float size, float count) { return min1(min1(a, b), op_pipe_c(a, b, size, count)); }

SDF_HD float op_staircase(float x, float stepval, float spread)
{
	return min1(stepval, frac1(x / spread) * spread) + floor1(x / spread) * stepval;
}
SDF_HD float op_smin(float a, float b, float k)
{
	float h = sat1(0.5f + 0.5f * (b - a) / k);
	return lerp1(b, a, h) - k * h * (1.f - h);
}
// the same with a literal k from the verified-divisor list (div_c): pass rk = 1.0f / k
SDF_HD float op_smin_c(float a, float b, float k, float rk)
{
	float h = sat1(0.5f + div_c(0.5f * (b - a), k, rk));
	return lerp1(b, a, h) - k * h * (1.f - h);
}
SDF_HD float op_smax2_c(float a, float b, float k, float rk)
{
	float h = sat1(0.5f - div_c(0.5f * (b - a), k, rk));
	return lerp1(b, a, h) + k * h * (1.f - h);
}
SDF_HD float op_smax1(float a, float b, float k)
{
	float h = sat1(0.5f - 0.5f * (b + a) / k);
	return lerp1(b, -a, h) + k * h * (1.f - h);
}
SDF_HD float op_smax2(float a, float b, float k)
{
	float h = sat1(0.5f - 0.5f * (b - a) / k);
	return lerp1(b, a, h) + k * h * (1.f - h);
}

// ---- colour helpers --------------------------------------------------------------------
SDF_HD vec3 hue_to_rgb(float H)
{
	float R = abs1(H * 6.f - 3.f) - 1.f;
	float G = 2.f - abs1(H * 6.f - 2.f);
	float B = 2.f - abs1(H * 6.f - 4.f);
	return saturate(V3(R, G, B));
}
SDF_HD vec3 hsv_to_rgb(vec3 hsv) { return ((hue_to_rgb(hsv.x) - 1.f) * hsv.y + 1.f) * hsv.z; }
SDF_HD float rgb_to_brightness(vec3 c) { return dot(c, V3(0.2126f, 0.7152f, 0.0722f)); }

// ---- checker floor with 4-tap footprint anti-aliasing (sdf_common.hlsl:24-60) -----------
SDF_HD vec4 checker_tap(vec3 p, vec3 dir)
{
	float to_move = p.y / dir.y;
	vec2 q = V2(p.x, p.z) - V2(dir.x, dir.z) * to_move;
	vec2 idx = floor(q);
	vec2 in_tile = q - idx;
	float parity = rne1(frac1((idx.x + idx.y) * 0.5f + 0.25f));
	float grey = parity > 0.5f ? 0.1f : 0.8f;
	vec2 border = 0.5f - abs(in_tile - 0.5f);
	return V4(grey, grey, grey, min1(border.x, border.y));
}
SDF_HD vec3 checker_color(vec3 p, vec3 dir, vec3 off_right, vec3 off_bottom)
{
	vec4 c1 = checker_tap(p, dir);
	vec4 c2 = checker_tap(p + off_right, dir);
	vec4 c3 = checker_tap(p + off_bottom, dir);
	vec4 c4 = checker_tap(p + off_bottom + off_right, dir);
	float total = c1.w + c2.w + c3.w + c4.w;
	vec3 sum = V3(c1.x, c1.y, c1.z) * c1.w + V3(c2.x, c2.y, c2.z) * c2.w + V3(c3.x, c3.y, c3.z) * c3.w + V3(c4.x, c4.y, c4.z) * c4.w;
	return sum / total;
}

// sky (sdf_common.hlsl:85-94) with the frame-uniform rotation sin/cos(-stime*0.025) passed in
SDF_HD vec3 sky_color(vec3 dir, float rot_s, float rot_c)
{
	vec2 r = rot2(V2(dir.x, dir.z), rot_s, rot_c);
	dir.x = r.x;
	dir.z = r.y;
	float n = turbulence3(dir * V3(1.f, 6.f, 1.f) * 2.5f);
	vec3 blue = V3(43.f, 164.f, 247.f) / 255.f;
	vec3 white = V3(212.f, 224.f, 238.f) / 255.f;
	vec3 sky = lerp(blue, white, n) * 1.2f;
	return lerp(V3s(0.25f), sky, sat1(dir.y * 8.f + 0.125f));
}
// the same sky with the cloud mix remapped to n * scale + bias (scenes with their own sky)
SDF_HD vec3 sky_color_mix(vec3 dir, float rot_s, float rot_c, float scale, float bias)
{
	vec2 r = rot2(V2(dir.x, dir.z), rot_s, rot_c);
	dir.x = r.x;
	dir.z = r.y;
	float n = turbulence3(dir * V3(1.f, 6.f, 1.f) * 2.5f);
	vec3 blue = V3(43.f, 164.f, 247.f) / 255.f;
	vec3 white = V3(212.f, 224.f, 238.f) / 255.f;
	vec3 sky = lerp(blue, white, n * scale + bias) * 1.2f;
	return lerp(V3s(0.25f), sky, sat1(dir.y * 8.f + 0.125f));
}

// ---- 2-D tilings used as materials (sdf_materials.hlsl:33-140) ------------------------------
SDF_HD uint32_t cell_hash_key(vec2 cell, float salt) { return (uint32_t)ftoi1(cell.x + cell.y * 217.743f + salt); }
SDF_HD vec2 voronoi_site(vec2 cell)
{
	float x = pcg_hashf((uint32_t)ftoi1(cell.x + cell.y * 217.743f));
	float y = pcg_hashf(cell_hash_key(cell, 2475.235f));
	return V2(x, y) * 2.f - 1.f;
}
// returns (id.x, id.y, distance to the nearest site, distance to the nearest cell edge)
SDF_HD vec4 voronoi(vec2 uv, float max_offset)
{
	const vec2 cell = floor(uv);
	const vec2 local = (uv - cell) - 0.5f;
	float best = 10.f, edge = 10.f;
	vec2 best_id = V2(0.f, 0.f), best_site = V2(0.f, 0.f);
	for (int x = -1; x < 2; ++x)
		for (int y = -1; y < 2; ++y)
		{
			const vec2 off = V2((float)x, (float)y);
			const vec2 id = cell + off;
			const vec2 site = off + voronoi_site(id) * max_offset;
			const float l = length(site - local);
			if (l < best) { best = l; best_id = id; best_site = site; }
		}
	for (int x = -1; x < 2; ++x)
		for (int y = -1; y < 2; ++y)
		{
			const vec2 off = V2((float)x, (float)y);
			const vec2 site = off + voronoi_site(cell + off) * max_offset;
			const vec2 mid = (site + best_site) * 0.5f;
			// for the nearest site itself this is 0 * inf = NaN, which min drops
			edge = min1(abs1(dot(normalize(best_site - mid), local - mid)), edge);
		}
	return V4(best_id.x, best_id.y, best, edge);
}
// quarter-circle Truchet tiles; returns (cell.x, cell.y, across-band u, along-band v) or miss_uv
SDF_HD vec4 truchet_band(vec2 uv, float chance, float width, vec2 miss_uv)
{
	const vec2 cell = floor(uv);
	vec2 local = (uv - cell) - 0.5f;
	const float flip3 = step1(frac1((cell.x + cell.y) * 0.5f + 0.25f), 0.5f) * 2.f - 1.f;
	const float flip2 = step1(pcg_hashf((uint32_t)ftoi1(cell.x + cell.y * 217.743f)), chance) * 2.f - 1.f;
	local.y = local.y * flip2;
	const float flip1 = step1(local.y, local.x) * 2.f - 1.f;
	local = local * flip1;
	local = local + V2(-0.5f, 0.5f);
	const float len = length(local);
	if (abs1(len - 0.5f) < width)
	{
		float a = (len - 0.5f + width) / (2.f * width);
		float b = atan21(local.y, -local.x) / (3.1415926f * 0.5f);
		a = lerp1(1.f - a, a, flip2 * flip3 * 0.5f + 0.5f);
		b = lerp1(1.f - b, b, flip3 * 0.5f + 0.5f);
		return V4(cell.x, cell.y, a, b);
	}
	return V4(cell.x, cell.y, miss_uv.x, miss_uv.y);
}
// woven bands; returns (cell.x, cell.y, position within the band) or miss_uv
SDF_HD vec4 braid(vec2 uv, float width, float run_length, float run_flip, vec2 miss_uv)
{
	const vec2 cell = floor(uv);
	vec2 local = (uv - cell) - 0.5f;
	const float t = frac1((cell.x + cell.y) / run_length) * run_length + 0.5f;
	const float flip = step1(t, run_flip);
	local = lerp(local, V2(local.y, local.x), flip);
	const vec2 rel = abs(local) / width;
	const vec2 over = V2(step1(1.f, rel.x), step1(1.f, rel.y));
	local = lerp(local, V2(local.y, local.x), over.x);
	local = lerp(local, miss_uv, over.x * over.y);
	return V4(cell.x, cell.y, local.x, local.y);
}

// ---- procedural materials (sdf_materials.hlsl:6-31) --------------------------------------
SDF_HD vec3 mat_marble(vec3 p, vec3 tint)
{
	float wave = dot(V3(3.f, 2.f, 1.f), p) * 2.f + turbulence3(p) * 5.f;
	float s = (1.f + sin1(wave)) * 0.5f;
	s = pow1(s, 0.5f);
	return tint * s;
}
SDF_HD vec3 mat_wood(vec3 p)
{
	float dist = sqrt1(p.x * p.x + p.y * p.y) + 0.125f * turbulence3(p);
	float s = 0.5f * abs1(sin1(2.f * 12.f * dist * 3.14159f));
	return V3(0.3125f + s, 0.117f + s, 0.117f);
}
SDF_HD vec4 mat_fire(vec3 p, float threshold)
{
	float turb = turbulence3(p) + 0.35f;
	turb = turb > threshold ? turb : 0.f;
	return V4(5.f * turb, 2.f * turb, 1.f * turb, 0.5f * turb);
}

// debug visualisations (sdf_materials.hlsl:143-186)
SDF_HD vec3 mat_debug_plane(float d)
{
	float ip;
	float fr = abs1(modf1(d, &ip)) * 1.2f;
	float band = modf1(ip / 5.f, &ip);
	vec3 band_color = band > 0.7f ? V3(1.f, 0.25f, 0.25f) : V3(0.75f, 0.75f, 1.f);
	fr = d < 25.f ? fr : 0.5f;
	vec3 col = fr < 1.f ? fr * fr * V3s(1.f) : band_color;
	col.y = d < 0.f ? (d > -0.01f ? 1.f : 0.f) : col.y;
	return col;
}
SDF_HD vec3 mat_iter_heat(uint32_t iter, uint32_t max_iter)
{
	float rel = (float)iter / (float)max_iter;
	if (rel < 0.1f) return lerp(V3(0.f, 0.f, 0.f), V3(0.f, 0.f, 1.f), rel / 0.1f);
	if (rel < 0.5f) return lerp(V3(0.f, 0.f, 1.f), V3(0.f, 1.f, 0.f), (rel - 0.1f) / 0.4f);
	if (rel < 0.9f) return lerp(V3(0.f, 1.f, 0.f), V3(1.f, 1.f, 0.f), (rel - 0.5f) / 0.4f);
	return lerp(V3(1.f, 1.f, 0.f), V3(1.f, 0.f, 0.f), (rel - 0.9f) / 0.1f);
}
SDF_HD float mat_coordinate_grid(vec3 p, vec3 n, float width)
{
	vec3 r = p - floor(p);
	r = abs(r - 0.5f);
	vec3 tick = saturate((r - 0.5f + width) * 100.f);
	vec3 mask = 1.f - abs(n);
	return dot(tick, mask);
}

// order (a, b, c) so that a >= b >= c, as three compare-exchanges (the fractal scenes' fold)
SDF_HD void sort3_desc(float &a, float &b, float &c)
{
	float t;
	if (c > b) { t = b; b = c; c = t; }
	if (b > a) { t = a; a = b; b = t; }
	if (c > b) { t = b; b = c; c = t; }
}

// ---- helpers every scene shares: checker floor, the standard sun (sdf_common.hlsl:62-94) ------
// needs SurfacePoint / Material / Light
// Ray-dependent part of the shared checker floor (sdf_common.hlsl:62-83 via
// sdf_primitives.hlsl:59-70): the fast plane divides by saturate(dot(dir, -n)) + 1e-20.
// The division height / denominator is made once per ray into a reciprocal and then done per
// step with div_c (sdfr_math.h): bit-identical to the IEEE divide for height = 0 and
// 2^-60 <= |height| <= 2^40 with any denominator in [1e-20, 2] (sdfr_selftest_math what = 3,
// swept over random and adversarial denominators in tests/test_gpu_math.py).
struct GroundInv { float denom, rdenom; };
SDF_HD GroundInv ground_setup(vec3 dir)
{
	GroundInv g;
	g.denom = sat1(dot(dir, -V3(0.f, 1.f, 0.f))) + 1e-20f;
	g.rdenom = rcp1(g.denom);
	return g;
}
SDF_HD float ground_dist(vec3 p, bool fast, const GroundInv &g)
{
	float d = dot(p, V3(0.f, 1.f, 0.f));
	return fast ? div_c(d, g.denom, g.rdenom) : d;
}
SDF_HD void ground_material(const FrameU &U, const SurfacePoint &sp, Material &m)
{
	if (on_surface(U, dot(sp.pos, V3(0.f, 1.f, 0.f))))
	{
		vec3 off_right = sp.right_off * sp.camera_distance;
		vec3 off_bottom = sp.bottom_off * sp.camera_distance;
		vec3 c = checker_color(sp.pos, sp.dir, off_right, off_bottom);
		m.diffuse = V4(c.x, c.y, c.z, 1.f);
		m.specular.x = m.specular.y = m.specular.z = 1.f;
	}
}
// the single white directional light all config scenes but light_shadows use
SDF_HD bool sun_light(int i, Light &L)
{
	if (i != 0) return false;
	L.pos = V3(-1.f, -1.f, 2.f);
	L.directional = true;
	L.color = V3(1.f, 1.f, 1.f);
	L.extend = 0.f;
	L.falloff = 0.f;
	return true;
}
SDF_HD void set_rgb(vec4 &c, float v) { c.x = v; c.y = v; c.z = v; }

} // namespace sdfr
