// sdfr_hlsl.h -- scenes written in the REFERENCE'S OWN DIALECT, compiled as they are.
//
// The plugin the reference exposes for the raymarch path is an .hlsl file that defines
//     void   map(GeometryInput geometry, MarchingInput march, MaterialInput material_input, inout MaterialOutput material_output,
//                bool geometry_step, inout float output_scene_distance)
//     void   map_normal(GeometryInput geometry, inout NormalOutput normal_output)
//     void   map_light(GeometryInput geometry, inout LightOutput output[LIGHT_COUNT], inout float ambient_lighting_factor)
//     float3 map_background(float3 dir, uint iter_count)
// with the OBJECT / OBJECT_TRANSPARENT / MATERIAL macros (pshader_sdf.hlsl:79-84; README.md:114-119; the 22 files under
// Engine/shader/scenes), textually substituted for "sdf_scene.hlsl" (Application.cpp:229,320).  sdfr_load_scene_hlsl takes
// such a file's text.  It is NOT parsed: a short textual pass (sdfr_hlsl.cpp: `inout` / `out` parameters become references,
// unsuffixed floating literals become floats, `static const` globals become members, [unroll]-style attributes and the
// #include lines of the library go) turns it into the body of a C++ class, and this header gives that class everything an
// HLSL scene expects to find:
//   * float2 / float3 / float4 with every swizzle (xyzw and rgba names, l-values included: `p.xz = opRepInf(p.xz, 2.f)`,
//     `v.yz = v.zy`, `color.rgb = 0.5f`), scalar <-> vector arithmetic, float3x3 + mul;
//   * the intrinsics with HLSL semantics and THIS library's arithmetic contract: round = half to even,
//     fmod = the FXC expansion, min / max = minNum / maxNum, dot / lerp / length / normalize / reflect / refract fused where
//     sdfr_math.h fuses them, sin cos atan2 exp2 log2 pow = the deterministic routines, tan = sin / cos, atan(x) = atan2(x, 1);
//   * the scene ABI types (sdf_structs.hlsl:4-130), material ids and macros (pshader_sdf.hlsl:67-81), the frame globals
//     (pshader_sdf.hlsl:17-36: eye, front_vec, right_vec, top_vec, stime, the five epsilons), pi / tau / sqrt_half / sqrt_two;
//   * the shader libraries under their HLSL names (sdf_primitives / sdf_ops / sdf_common / sdf_materials / noise: sdSphere ...
//     turbulence), each a thin wrapper of the function the built-in scenes use (sdfr_lib.h, sdfr_noise.h) -- so a scene
//     loaded this way and the same scene compiled ahead of time render the same bits.  All three snoise overloads,
//     grad4, mod289 and permute of noise.hlsl are there.
// SceneAdapter<UserScene> then presents the class to the pixel kernel as any other scene (dist / material / normal / lights /
// background).  A geometry step sees the GeometryInput the reference hands it: the sample's running camera_distance and the
// pixel's ray offsets (SceneReadsMarchState, sdfr_pixel.h).
#pragma once
#include "sdfr_pixel.h"
#include "sdfr_hlsl_swizzles.h"

namespace sdfr {
namespace hlsl {

typedef unsigned int uint;
struct float2;
struct float3;
struct float4;

// ---- swizzle proxies: empty types that live in a union with the vector's components ---------------------
template <int A, int B>
struct swz2
{
	SDF_HD float *p() { return reinterpret_cast<float *>(this); }
	SDF_HD const float *p() const { return reinterpret_cast<const float *>(this); }
	SDF_HD operator float2() const;
	SDF_HD swz2 &operator=(const float2 &v);
	SDF_HD swz2 &operator=(const swz2 &o);
	template <int C, int D> SDF_HD swz2 &operator=(const swz2<C, D> &o);
	SDF_HD swz2 &operator=(float s);
	SDF_HD swz2 &operator+=(const float2 &v);
	SDF_HD swz2 &operator-=(const float2 &v);
	SDF_HD swz2 &operator*=(const float2 &v);
	SDF_HD swz2 &operator/=(const float2 &v);
	SDF_HD swz2 &operator*=(float s);
	SDF_HD swz2 &operator/=(float s);
};
template <int A, int B, int C>
struct swz3
{
	SDF_HD float *p() { return reinterpret_cast<float *>(this); }
	SDF_HD const float *p() const { return reinterpret_cast<const float *>(this); }
	SDF_HD operator float3() const;
	SDF_HD swz3 &operator=(const float3 &v);
	SDF_HD swz3 &operator=(const swz3 &o);
	template <int D, int E, int F> SDF_HD swz3 &operator=(const swz3<D, E, F> &o);
	SDF_HD swz3 &operator=(float s);
	SDF_HD swz3 &operator+=(const float3 &v);
	SDF_HD swz3 &operator-=(const float3 &v);
	SDF_HD swz3 &operator*=(const float3 &v);
	SDF_HD swz3 &operator/=(const float3 &v);
	SDF_HD swz3 &operator*=(float s);
	SDF_HD swz3 &operator/=(float s);
};
template <int A, int B, int C, int D>
struct swz4
{
	SDF_HD float *p() { return reinterpret_cast<float *>(this); }
	SDF_HD const float *p() const { return reinterpret_cast<const float *>(this); }
	SDF_HD operator float4() const;
	SDF_HD swz4 &operator=(const float4 &v);
	SDF_HD swz4 &operator=(const swz4 &o);
	SDF_HD swz4 &operator=(float s);
};

struct float2
{
	union
	{
		struct { float x, y; };
		struct { float r, g; };
		SDFR_HLSL_SWIZZLES_2
	};
	SDF_HD float2() {}
	SDF_HD float2(float s) : x(s), y(s) {}
	SDF_HD float2(float x_, float y_) : x(x_), y(y_) {}
	SDF_HD float2(const float2 &o) : x(o.x), y(o.y) {}
	SDF_HD float2(vec2 v) : x(v.x), y(v.y) {}
	SDF_HD float2 &operator=(const float2 &o) { x = o.x; y = o.y; return *this; }
	SDF_HD operator vec2() const { return V2(x, y); }
};
struct float3
{
	union
	{
		struct { float x, y, z; };
		struct { float r, g, b; };
		SDFR_HLSL_SWIZZLES_3
	};
	SDF_HD float3() {}
	SDF_HD float3(float s) : x(s), y(s), z(s) {}
	SDF_HD float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
	SDF_HD float3(const float2 &xy_, float z_) : x(xy_.x), y(xy_.y), z(z_) {}
	SDF_HD float3(float x_, const float2 &yz_) : x(x_), y(yz_.x), z(yz_.y) {}
	SDF_HD float3(const float3 &o) : x(o.x), y(o.y), z(o.z) {}
	SDF_HD float3(vec3 v) : x(v.x), y(v.y), z(v.z) {}
	SDF_HD float3 &operator=(const float3 &o) { x = o.x; y = o.y; z = o.z; return *this; }
	SDF_HD operator vec3() const { return V3(x, y, z); }
};
struct float4
{
	union
	{
		struct { float x, y, z, w; };
		struct { float r, g, b, a; };
		SDFR_HLSL_SWIZZLES_4
	};
	SDF_HD float4() {}
	SDF_HD float4(float s) : x(s), y(s), z(s), w(s) {}
	SDF_HD float4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
	SDF_HD float4(const float3 &v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
	SDF_HD float4(float x_, const float3 &v) : x(x_), y(v.x), z(v.y), w(v.z) {}
	SDF_HD float4(const float2 &a_, const float2 &b_) : x(a_.x), y(a_.y), z(b_.x), w(b_.y) {}
	SDF_HD float4(const float2 &a_, float z_, float w_) : x(a_.x), y(a_.y), z(z_), w(w_) {}
	SDF_HD float4(const float4 &o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
	SDF_HD float4(vec4 v) : x(v.x), y(v.y), z(v.z), w(v.w) {}
	SDF_HD float4 &operator=(const float4 &o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
	SDF_HD operator vec4() const { return V4(x, y, z, w); }
};
struct float3x3 { float m[3][3]; }; // row-major, brace-initialised like HLSL's { m00, m01, ... }

// proxies: reads make a temporary first, so `v.yz = v.zy` is a swap as in HLSL (copy-in / copy-out)
template <int A, int B> SDF_HD swz2<A, B>::operator float2() const { return float2(p()[A], p()[B]); }
template <int A, int B> SDF_HD swz2<A, B> &swz2<A, B>::operator=(const float2 &v) { const float a = v.x, b = v.y; p()[A] = a; p()[B] = b; return *this; }
template <int A, int B> SDF_HD swz2<A, B> &swz2<A, B>::operator=(const swz2 &o) { const float2 t = o; return *this = t; }
template <int A, int B> template <int C, int D> SDF_HD swz2<A, B> &swz2<A, B>::operator=(const swz2<C, D> &o) { const float2 t = o; return *this = t; }
template <int A, int B> SDF_HD swz2<A, B> &swz2<A, B>::operator=(float s) { p()[A] = s; p()[B] = s; return *this; }
template <int A, int B> SDF_HD swz2<A, B> &swz2<A, B>::operator+=(const float2 &v) { const float2 t = *this; return *this = float2(t.x + v.x, t.y + v.y); }
template <int A, int B> SDF_HD swz2<A, B> &swz2<A, B>::operator-=(const float2 &v) { const float2 t = *this; return *this = float2(t.x - v.x, t.y - v.y); }
template <int A, int B> SDF_HD swz2<A, B> &swz2<A, B>::operator*=(const float2 &v) { const float2 t = *this; return *this = float2(t.x * v.x, t.y * v.y); }
template <int A, int B> SDF_HD swz2<A, B> &swz2<A, B>::operator/=(const float2 &v) { const float2 t = *this; return *this = float2(t.x / v.x, t.y / v.y); }
template <int A, int B> SDF_HD swz2<A, B> &swz2<A, B>::operator*=(float s) { const float2 t = *this; return *this = float2(t.x * s, t.y * s); }
template <int A, int B> SDF_HD swz2<A, B> &swz2<A, B>::operator/=(float s) { const float2 t = *this; return *this = float2(t.x / s, t.y / s); }
template <int A, int B, int C> SDF_HD swz3<A, B, C>::operator float3() const { return float3(p()[A], p()[B], p()[C]); }
template <int A, int B, int C> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator=(const float3 &v) { const float a = v.x, b = v.y, c = v.z; p()[A] = a; p()[B] = b; p()[C] = c; return *this; }
template <int A, int B, int C> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator=(const swz3 &o) { const float3 t = o; return *this = t; }
template <int A, int B, int C> template <int D, int E, int F> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator=(const swz3<D, E, F> &o) { const float3 t = o; return *this = t; }
template <int A, int B, int C> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator=(float s) { p()[A] = s; p()[B] = s; p()[C] = s; return *this; }
template <int A, int B, int C> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator+=(const float3 &v) { const float3 t = *this; return *this = float3(t.x + v.x, t.y + v.y, t.z + v.z); }
template <int A, int B, int C> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator-=(const float3 &v) { const float3 t = *this; return *this = float3(t.x - v.x, t.y - v.y, t.z - v.z); }
template <int A, int B, int C> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator*=(const float3 &v) { const float3 t = *this; return *this = float3(t.x * v.x, t.y * v.y, t.z * v.z); }
template <int A, int B, int C> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator/=(const float3 &v) { const float3 t = *this; return *this = float3(t.x / v.x, t.y / v.y, t.z / v.z); }
template <int A, int B, int C> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator*=(float s) { const float3 t = *this; return *this = float3(t.x * s, t.y * s, t.z * s); }
template <int A, int B, int C> SDF_HD swz3<A, B, C> &swz3<A, B, C>::operator/=(float s) { const float3 t = *this; return *this = float3(t.x / s, t.y / s, t.z / s); }
template <int A, int B, int C, int D> SDF_HD swz4<A, B, C, D>::operator float4() const { return float4(p()[A], p()[B], p()[C], p()[D]); }
template <int A, int B, int C, int D> SDF_HD swz4<A, B, C, D> &swz4<A, B, C, D>::operator=(const float4 &v)
{
	const float a = v.x, b = v.y, c = v.z, d = v.w;
	p()[A] = a; p()[B] = b; p()[C] = c; p()[D] = d;
	return *this;
}
template <int A, int B, int C, int D> SDF_HD swz4<A, B, C, D> &swz4<A, B, C, D>::operator=(const swz4 &o) { const float4 t = o; return *this = t; }
template <int A, int B, int C, int D> SDF_HD swz4<A, B, C, D> &swz4<A, B, C, D>::operator=(float s) { p()[A] = s; p()[B] = s; p()[C] = s; p()[D] = s; return *this; }

// ---- arithmetic: one IEEE operation per component, in source order (nothing fuses) ------------------------
SDF_HD float2 operator+(float2 a, float2 b) { return float2(a.x + b.x, a.y + b.y); }
SDF_HD float2 operator-(float2 a, float2 b) { return float2(a.x - b.x, a.y - b.y); }
SDF_HD float2 operator*(float2 a, float2 b) { return float2(a.x * b.x, a.y * b.y); }
SDF_HD float2 operator/(float2 a, float2 b) { return float2(a.x / b.x, a.y / b.y); }
SDF_HD float2 operator+(float2 a, float b) { return float2(a.x + b, a.y + b); }
SDF_HD float2 operator-(float2 a, float b) { return float2(a.x - b, a.y - b); }
SDF_HD float2 operator*(float2 a, float b) { return float2(a.x * b, a.y * b); }
SDF_HD float2 operator/(float2 a, float b) { return float2(a.x / b, a.y / b); }
SDF_HD float2 operator+(float a, float2 b) { return float2(a + b.x, a + b.y); }
SDF_HD float2 operator-(float a, float2 b) { return float2(a - b.x, a - b.y); }
SDF_HD float2 operator*(float a, float2 b) { return float2(a * b.x, a * b.y); }
SDF_HD float2 operator/(float a, float2 b) { return float2(a / b.x, a / b.y); }
SDF_HD float2 operator-(float2 a) { return float2(-a.x, -a.y); }
SDF_HD float3 operator+(float3 a, float3 b) { return float3(a.x + b.x, a.y + b.y, a.z + b.z); }
SDF_HD float3 operator-(float3 a, float3 b) { return float3(a.x - b.x, a.y - b.y, a.z - b.z); }
SDF_HD float3 operator*(float3 a, float3 b) { return float3(a.x * b.x, a.y * b.y, a.z * b.z); }
SDF_HD float3 operator/(float3 a, float3 b) { return float3(a.x / b.x, a.y / b.y, a.z / b.z); }
SDF_HD float3 operator+(float3 a, float b) { return float3(a.x + b, a.y + b, a.z + b); }
SDF_HD float3 operator-(float3 a, float b) { return float3(a.x - b, a.y - b, a.z - b); }
SDF_HD float3 operator*(float3 a, float b) { return float3(a.x * b, a.y * b, a.z * b); }
SDF_HD float3 operator/(float3 a, float b) { return float3(a.x / b, a.y / b, a.z / b); }
SDF_HD float3 operator+(float a, float3 b) { return float3(a + b.x, a + b.y, a + b.z); }
SDF_HD float3 operator-(float a, float3 b) { return float3(a - b.x, a - b.y, a - b.z); }
SDF_HD float3 operator*(float a, float3 b) { return float3(a * b.x, a * b.y, a * b.z); }
SDF_HD float3 operator/(float a, float3 b) { return float3(a / b.x, a / b.y, a / b.z); }
SDF_HD float3 operator-(float3 a) { return float3(-a.x, -a.y, -a.z); }
SDF_HD float4 operator+(float4 a, float4 b) { return float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
SDF_HD float4 operator-(float4 a, float4 b) { return float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
SDF_HD float4 operator*(float4 a, float4 b) { return float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
SDF_HD float4 operator/(float4 a, float4 b) { return float4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
SDF_HD float4 operator+(float4 a, float b) { return float4(a.x + b, a.y + b, a.z + b, a.w + b); }
SDF_HD float4 operator-(float4 a, float b) { return float4(a.x - b, a.y - b, a.z - b, a.w - b); }
SDF_HD float4 operator*(float4 a, float b) { return float4(a.x * b, a.y * b, a.z * b, a.w * b); }
SDF_HD float4 operator/(float4 a, float b) { return float4(a.x / b, a.y / b, a.z / b, a.w / b); }
SDF_HD float4 operator+(float a, float4 b) { return float4(a + b.x, a + b.y, a + b.z, a + b.w); }
SDF_HD float4 operator-(float a, float4 b) { return float4(a - b.x, a - b.y, a - b.z, a - b.w); }
SDF_HD float4 operator*(float a, float4 b) { return float4(a * b.x, a * b.y, a * b.z, a * b.w); }
SDF_HD float4 operator/(float a, float4 b) { return float4(a / b.x, a / b.y, a / b.z, a / b.w); }
SDF_HD float4 operator-(float4 a) { return float4(-a.x, -a.y, -a.z, -a.w); }
#define SDFR_HLSL_COMPOUND(T) \
	SDF_HD T &operator+=(T &a, T b) { a = a + b; return a; } \
	SDF_HD T &operator-=(T &a, T b) { a = a - b; return a; } \
	SDF_HD T &operator*=(T &a, T b) { a = a * b; return a; } \
	SDF_HD T &operator/=(T &a, T b) { a = a / b; return a; } \
	SDF_HD T &operator+=(T &a, float b) { a = a + b; return a; } \
	SDF_HD T &operator-=(T &a, float b) { a = a - b; return a; } \
	SDF_HD T &operator*=(T &a, float b) { a = a * b; return a; } \
	SDF_HD T &operator/=(T &a, float b) { a = a / b; return a; }
SDFR_HLSL_COMPOUND(float2)
SDFR_HLSL_COMPOUND(float3)
SDFR_HLSL_COMPOUND(float4)

// ---- intrinsics ---------------------------------------------------------------------------------------------
// scalar forms (float; the integer forms HLSL code reaches for now and then)
SDF_HD float abs(float a) { return abs1(a); }
SDF_HD int abs(int a) { return a < 0 ? -a : a; }
SDF_HD float min(float a, float b) { return min1(a, b); }
SDF_HD float max(float a, float b) { return max1(a, b); }
SDF_HD int min(int a, int b) { return a < b ? a : b; }
SDF_HD int max(int a, int b) { return a > b ? a : b; }
SDF_HD uint min(uint a, uint b) { return a < b ? a : b; }
SDF_HD uint max(uint a, uint b) { return a > b ? a : b; }
// an integer beside a float is a float (`max(y - pen, 0)`)
SDF_HD float min(float a, int b) { return min1(a, (float)b); }
SDF_HD float min(int a, float b) { return min1((float)a, b); }
SDF_HD float max(float a, int b) { return max1(a, (float)b); }
SDF_HD float max(int a, float b) { return max1((float)a, b); }
SDF_HD float min(float a, uint b) { return min1(a, (float)b); }
SDF_HD float min(uint a, float b) { return min1((float)a, b); }
SDF_HD float max(float a, uint b) { return max1(a, (float)b); }
SDF_HD float max(uint a, float b) { return max1((float)a, b); }
SDF_HD float floor(float a) { return floor1(a); }
SDF_HD float ceil(float a) { return -floor1(-a); }
SDF_HD float trunc(float a) { return trunc1(a); }
SDF_HD float round(float a) { return rne1(a); } // half to even
SDF_HD float frac(float a) { return frac1(a); }
SDF_HD float saturate(float a) { return sat1(a); }
SDF_HD float clamp(float a, float lo, float hi) { return clamp1(a, lo, hi); }
SDF_HD float lerp(float a, float b, float t) { return lerp1(a, b, t); }
SDF_HD float step(float edge, float x) { return step1(edge, x); }
SDF_HD float sign(float a) { return sign1(a); }
SDF_HD float sqrt(float a) { return sqrt1(a); }
SDF_HD float rsqrt(float a) { return rsqrt1(a); }
SDF_HD float rcp(float a) { return rcp1(a); }
SDF_HD float sin(float a) { return sin1(a); }
SDF_HD float cos(float a) { return cos1(a); }
SDF_HD float tan(float a) { return sin1(a) / cos1(a); }       // what the HLSL compiler expands tan to
SDF_HD float atan2(float y, float x) { return atan21(y, x); }
SDF_HD float atan(float a) { return atan21(a, 1.f); }
SDF_HD float exp2(float a) { return exp21(a); }
SDF_HD float log2(float a) { return log21(a); }
SDF_HD float pow(float x, float y) { return pow1(x, y); }     // exp2(y * log2(x))
SDF_HD float exp(float a) { return exp21(a * 1.44269504088896341f); }
SDF_HD float log(float a) { return log21(a) * 0.69314718055994531f; }
SDF_HD float fmod(float a, float b) { return fmod1(a, b); }
SDF_HD float modf(float a, float &ip) { return modf1(a, &ip); }
SDF_HD float mad(float a, float b, float c) { return fma1(a, b, c); }
SDF_HD void sincos(float a, float &s, float &c) { s = sin1(a); c = cos1(a); }
SDF_HD float smoothstep(float lo, float hi, float x) { const float t = sat1((x - lo) / (hi - lo)); return t * t * (3.f - 2.f * t); }
SDF_HD bool any(float a) { return a != 0.f; }
SDF_HD bool all(float a) { return a != 0.f; }
SDF_HD float radians(float a) { return a * 0.01745329251994329577f; }
SDF_HD float degrees(float a) { return a * 57.2957795130823208768f; }
// D3D casts of a float to an integer saturate (a C++ cast out of range is undefined): `(int)(expr)` becomes ftoi_(expr)
SDF_HD int ftoi_(float a) { return ftoi1(a); }
SDF_HD int ftoi_(int a) { return a; }
SDF_HD int ftoi_(uint a) { return (int)a; }
SDF_HD uint ftou_(float a) { return a >= 4294967296.f ? 0xffffffffu : (a > 0.f ? (uint)a : 0u); }
SDF_HD uint ftou_(int a) { return (uint)a; }
SDF_HD uint ftou_(uint a) { return a; }

// component-wise forms
#define SDFR_HLSL_MAP1(F) \
	SDF_HD float2 F(float2 a) { return float2(F(a.x), F(a.y)); } \
	SDF_HD float3 F(float3 a) { return float3(F(a.x), F(a.y), F(a.z)); } \
	SDF_HD float4 F(float4 a) { return float4(F(a.x), F(a.y), F(a.z), F(a.w)); }
SDFR_HLSL_MAP1(abs) SDFR_HLSL_MAP1(floor) SDFR_HLSL_MAP1(ceil) SDFR_HLSL_MAP1(trunc) SDFR_HLSL_MAP1(round) SDFR_HLSL_MAP1(frac) SDFR_HLSL_MAP1(saturate)
SDFR_HLSL_MAP1(sign) SDFR_HLSL_MAP1(sqrt) SDFR_HLSL_MAP1(rsqrt) SDFR_HLSL_MAP1(sin) SDFR_HLSL_MAP1(cos) SDFR_HLSL_MAP1(tan) SDFR_HLSL_MAP1(atan)
SDFR_HLSL_MAP1(exp2) SDFR_HLSL_MAP1(log2) SDFR_HLSL_MAP1(exp) SDFR_HLSL_MAP1(log)
#define SDFR_HLSL_MAP2(F) \
	SDF_HD float2 F(float2 a, float2 b) { return float2(F(a.x, b.x), F(a.y, b.y)); } \
	SDF_HD float3 F(float3 a, float3 b) { return float3(F(a.x, b.x), F(a.y, b.y), F(a.z, b.z)); } \
	SDF_HD float4 F(float4 a, float4 b) { return float4(F(a.x, b.x), F(a.y, b.y), F(a.z, b.z), F(a.w, b.w)); } \
	SDF_HD float2 F(float2 a, float b) { return float2(F(a.x, b), F(a.y, b)); } \
	SDF_HD float3 F(float3 a, float b) { return float3(F(a.x, b), F(a.y, b), F(a.z, b)); } \
	SDF_HD float4 F(float4 a, float b) { return float4(F(a.x, b), F(a.y, b), F(a.z, b), F(a.w, b)); } \
	SDF_HD float2 F(float a, float2 b) { return float2(F(a, b.x), F(a, b.y)); } \
	SDF_HD float3 F(float a, float3 b) { return float3(F(a, b.x), F(a, b.y), F(a, b.z)); } \
	SDF_HD float4 F(float a, float4 b) { return float4(F(a, b.x), F(a, b.y), F(a, b.z), F(a, b.w)); }
SDFR_HLSL_MAP2(min) SDFR_HLSL_MAP2(max) SDFR_HLSL_MAP2(step) SDFR_HLSL_MAP2(pow) SDFR_HLSL_MAP2(fmod) SDFR_HLSL_MAP2(atan2)
SDF_HD float2 clamp(float2 a, float2 lo, float2 hi) { return float2(clamp1(a.x, lo.x, hi.x), clamp1(a.y, lo.y, hi.y)); }
SDF_HD float3 clamp(float3 a, float3 lo, float3 hi) { return float3(clamp1(a.x, lo.x, hi.x), clamp1(a.y, lo.y, hi.y), clamp1(a.z, lo.z, hi.z)); }
SDF_HD float4 clamp(float4 a, float4 lo, float4 hi) { return float4(clamp1(a.x, lo.x, hi.x), clamp1(a.y, lo.y, hi.y), clamp1(a.z, lo.z, hi.z), clamp1(a.w, lo.w, hi.w)); }
SDF_HD float2 clamp(float2 a, float lo, float hi) { return clamp(a, float2(lo), float2(hi)); }
SDF_HD float3 clamp(float3 a, float lo, float hi) { return clamp(a, float3(lo), float3(hi)); }
SDF_HD float4 clamp(float4 a, float lo, float hi) { return clamp(a, float4(lo), float4(hi)); }
SDF_HD float2 lerp(float2 a, float2 b, float t) { return float2(lerp1(a.x, b.x, t), lerp1(a.y, b.y, t)); }
SDF_HD float3 lerp(float3 a, float3 b, float t) { return float3(lerp1(a.x, b.x, t), lerp1(a.y, b.y, t), lerp1(a.z, b.z, t)); }
SDF_HD float4 lerp(float4 a, float4 b, float t) { return float4(lerp1(a.x, b.x, t), lerp1(a.y, b.y, t), lerp1(a.z, b.z, t), lerp1(a.w, b.w, t)); }
SDF_HD float2 lerp(float2 a, float2 b, float2 t) { return float2(lerp1(a.x, b.x, t.x), lerp1(a.y, b.y, t.y)); }
SDF_HD float3 lerp(float3 a, float3 b, float3 t) { return float3(lerp1(a.x, b.x, t.x), lerp1(a.y, b.y, t.y), lerp1(a.z, b.z, t.z)); }
SDF_HD float4 lerp(float4 a, float4 b, float4 t) { return float4(lerp1(a.x, b.x, t.x), lerp1(a.y, b.y, t.y), lerp1(a.z, b.z, t.z), lerp1(a.w, b.w, t.w)); }
SDF_HD float dot(float2 a, float2 b) { return fma1(a.y, b.y, a.x * b.x); }
SDF_HD float dot(float3 a, float3 b) { return fma1(a.z, b.z, fma1(a.y, b.y, a.x * b.x)); }
SDF_HD float dot(float4 a, float4 b) { return fma1(a.w, b.w, fma1(a.z, b.z, fma1(a.y, b.y, a.x * b.x))); }
SDF_HD float length(float a) { return abs1(a); }
SDF_HD float length(float2 a) { return sqrt1(dot(a, a)); }
SDF_HD float length(float3 a) { return sqrt1(dot(a, a)); }
SDF_HD float length(float4 a) { return sqrt1(dot(a, a)); }
SDF_HD float distance(float2 a, float2 b) { return length(a - b); }
SDF_HD float distance(float3 a, float3 b) { return length(a - b); }
SDF_HD float2 normalize(float2 a) { return a * rsqrt1(dot(a, a)); }
SDF_HD float3 normalize(float3 a) { return a * rsqrt1(dot(a, a)); }
SDF_HD float4 normalize(float4 a) { return a * rsqrt1(dot(a, a)); }
SDF_HD float3 cross(float3 a, float3 b) { return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
SDF_HD float3 reflect(float3 i, float3 n) { return float3(sdfr::reflect((vec3)i, (vec3)n)); }
SDF_HD float3 refract(float3 i, float3 n, float eta) { return float3(sdfr::refract((vec3)i, (vec3)n, eta)); }
SDF_HD float3 mad(float3 a, float3 b, float3 c) { return float3(fma1(a.x, b.x, c.x), fma1(a.y, b.y, c.y), fma1(a.z, b.z, c.z)); }
SDF_HD bool any(float2 a) { return a.x != 0.f || a.y != 0.f; }
SDF_HD bool any(float3 a) { return a.x != 0.f || a.y != 0.f || a.z != 0.f; }
SDF_HD bool any(float4 a) { return a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f; }
SDF_HD bool all(float2 a) { return a.x != 0.f && a.y != 0.f; }
SDF_HD bool all(float3 a) { return a.x != 0.f && a.y != 0.f && a.z != 0.f; }
SDF_HD bool all(float4 a) { return a.x != 0.f && a.y != 0.f && a.z != 0.f && a.w != 0.f; }
// mul(matrix, column vector): a dot product per row
SDF_HD float3 mul(const float3x3 &m, float3 v)
{
	return float3(dot(float3(m.m[0][0], m.m[0][1], m.m[0][2]), v), dot(float3(m.m[1][0], m.m[1][1], m.m[1][2]), v), dot(float3(m.m[2][0], m.m[2][1], m.m[2][2]), v));
}
SDF_HD float3 mul(float3 v, const float3x3 &m)
{
	return float3(dot(v, float3(m.m[0][0], m.m[1][0], m.m[2][0])), dot(v, float3(m.m[0][1], m.m[1][1], m.m[2][1])), dot(v, float3(m.m[0][2], m.m[1][2], m.m[2][2])));
}

// ---- the scene ABI (sdf_structs.hlsl:4-130) ---------------------------------------------------------------
struct GeometryInput
{
	float3 pos;               // current position
	float4 dir;               // ray direction; w = 1: fast (analytic) primitives allowed, 0: exact distances wanted
	float camera_distance;
	float3 right_ray_offset;  // pixel footprint per unit of distance
	float3 bottom_ray_offset;
};
struct MarchingInput
{
	bool is_inside;
	float3 last_transparent_pos;
	bool has_transparent;
	bool is_shadow_pass;
};
struct NormalOutput
{
	float normal_sample_dist;
	float3 normal;
	bool use_normal;
};
struct MaterialInput
{
	float3 obj_normal;
	uint iteration_count;
	float scene_distance;
};
struct MaterialOutput
{
	uint material_id;
	float4 material_position;
	float4 material_properties;
	float4 diffuse_color;   // rgb + alpha
	float4 specular_color;  // rgb + power
	float3 emissive_color;
	float3 reflection_color;
	float3 refraction_color;
	float optical_index;
	float optical_density;
	float4 normal;          // xyz + blend
	uint max_cost;
	bool use_hdr;
};
struct LightOutput
{
	bool used;
	float4 pos;             // w = 1: directional
	float extend;
	float3 color;
	float falloff;
};

// driver constants a scene may name (pshader_sdf.hlsl:53-76)
enum { LIGHT_COUNT = SDFR_MAX_LIGHTS };
enum
{
	MATERIAL_NONE = MAT_NONE, MATERIAL_PLAIN = MAT_PLAIN, MATERIAL_ITER = MAT_ITER, MATERIAL_NORMAL1 = MAT_NORMAL1, MATERIAL_NORMAL2 = MAT_NORMAL2,
	MATERIAL_DISTANCE_PLANE = MAT_DISTANCE_PLANE, MATERIAL_WOOD = MAT_WOOD, MATERIAL_MARBLE_DARK = MAT_MARBLE_DARK, MATERIAL_MARBLE_LIGHT = MAT_MARBLE_LIGHT,
	MATERIAL_FIRE = MAT_FIRE
};
// the macros of the plugin ABI (pshader_sdf.hlsl:79-81)
#define OBJECT(distance) output_scene_distance = min(output_scene_distance, distance)
#define OBJECT_TRANSPARENT(distance, distance_transparent) \
	output_scene_distance = ((march.has_transparent && (distance_transparent) < dist_eps) ? output_scene_distance : min(output_scene_distance, distance))
#define MATERIAL(distance) (abs(distance) < dist_eps)

// the members every scene class starts with: the frame globals of pshader_sdf.hlsl:17-36 and math_constants.hlsl
#define SDFR_HLSL_FRAME_MEMBERS(CLASS) \
	const FrameU &U; \
	const float3 eye, front_vec, right_vec, top_vec; \
	const float stime; \
	const float dist_eps, grad_eps, reflect_eps, refract_eps, shadow_eps; \
	const float sqrt_half = SDFR_SQRT_HALF, sqrt_two = SDFR_SQRT_TWO, pi = SDFR_PI, tau = SDFR_TAU; \
	SDF_HD explicit CLASS(const FrameU &u) \
		: U(u), eye(u.eye), front_vec(u.front), right_vec(u.right), top_vec(u.top), stime(u.stime), dist_eps(u.dist_eps), grad_eps(u.grad_eps), \
		  reflect_eps(u.reflect_eps), refract_eps(u.refract_eps), shadow_eps(u.shadow_eps) {}

// ---- a scene class as the pixel kernel's Scene ----------------------------------------------------------------
template <class S>
struct SceneAdapter
{
	static SDF_HD void prepare(FrameU &) {}
	struct RayInv { RayFlags flags; };
	static SDF_HD RayInv ray_setup(const FrameU &, vec3, const RayFlags &f)
	{
		RayInv r;
		r.flags = f;
		return r;
	}
	static SDF_HD GeometryInput geometry_of(const SurfacePoint &sp)
	{
		GeometryInput g;
		g.pos = float3(sp.pos);
		g.dir = float4(float3(sp.dir), 0.f);
		g.camera_distance = sp.camera_distance;
		g.right_ray_offset = float3(sp.right_off);
		g.bottom_ray_offset = float3(sp.bottom_off);
		return g;
	}
	static SDF_HD MaterialOutput zero_material()
	{
		MaterialOutput m;
		m.material_id = 0u;
		m.material_position = float4(0.f);
		m.material_properties = float4(0.f);
		m.diffuse_color = float4(0.f);
		m.specular_color = float4(0.f);
		m.emissive_color = float3(0.f);
		m.reflection_color = float3(0.f);
		m.refraction_color = float3(0.f);
		m.optical_index = 0.f;
		m.optical_density = 0.f;
		m.normal = float4(0.f);
		m.max_cost = 0u;
		m.use_hdr = false;
		return m;
	}
	// map(..., geometry_step = true), as map_geometry calls it (pshader_sdf.hlsl:111-135), with the GeometryInput the reference
	// hands it: the running camera_distance of the sample and the pixel's ray offsets (:187-218, 297-302); the three samples
	// of the normal carry the hit's (:164-177)
	static constexpr bool geometry_reads_march_state = true;
	static SDF_HD float dist(const FrameU &U, const RayInv &R, vec3 p, vec3 dir, bool fast, const GeoStep &gs)
	{
		S scene(U);
		GeometryInput g;
		g.pos = float3(p);
		g.dir = float4(float3(dir), fast ? 1.f : 0.f);
		g.camera_distance = gs.camera_distance;
		g.right_ray_offset = float3(gs.right_off);
		g.bottom_ray_offset = float3(gs.bottom_off);
		MarchingInput march;
		march.is_inside = false;
		march.last_transparent_pos = float3(R.flags.last_transparent_pos);
		march.has_transparent = R.flags.has_transparent;
		march.is_shadow_pass = R.flags.is_shadow;
		MaterialInput mi;
		mi.obj_normal = float3(0.f);
		mi.iteration_count = 0u;
		mi.scene_distance = 0.f;
		MaterialOutput mo = zero_material();
		float d = 3e38f;
		scene.map(g, march, mi, mo, true, d);
		return d;
	}
	// map(..., geometry_step = false) with a zeroed MarchingInput, as map_material calls it (pshader_sdf.hlsl:137-162)
	static SDF_HD void material(const FrameU &U, const SurfacePoint &sp, Material &m)
	{
		S scene(U);
		const GeometryInput g = geometry_of(sp);
		MarchingInput march;
		march.is_inside = false;
		march.last_transparent_pos = float3(0.f);
		march.has_transparent = false;
		march.is_shadow_pass = false;
		MaterialInput mi;
		mi.obj_normal = float3(sp.normal);
		mi.iteration_count = sp.iteration_count;
		mi.scene_distance = sp.scene_distance;
		MaterialOutput mo;
		mo.material_id = m.id;
		mo.material_position = float4(float3(m.mpos), 0.f);
		mo.material_properties = float4(m.prop_x, 0.f, 0.f, 0.f);
		mo.diffuse_color = float4(m.diffuse);
		mo.specular_color = float4(m.specular);
		mo.emissive_color = float3(m.emissive);
		mo.reflection_color = float3(m.reflection);
		mo.refraction_color = float3(m.refraction);
		mo.optical_index = m.ior;
		mo.optical_density = 0.f;
		mo.normal = float4(m.normal);
		mo.max_cost = m.max_cost;
		mo.use_hdr = m.use_hdr;
		float d = 3e38f;
		scene.map(g, march, mi, mo, false, d);
		m.id = mo.material_id;
		m.mpos = V3(mo.material_position.x, mo.material_position.y, mo.material_position.z);
		m.prop_x = mo.material_properties.x;
		m.diffuse = mo.diffuse_color;
		m.specular = mo.specular_color;
		m.emissive = mo.emissive_color;
		m.reflection = mo.reflection_color;
		m.refraction = mo.refraction_color;
		m.ior = mo.optical_index;
		m.normal = mo.normal;
		m.max_cost = mo.max_cost;
		m.use_hdr = mo.use_hdr;
	}
	// map_normal (pshader_sdf.hlsl:318-330)
	static SDF_HD void normal(const FrameU &U, const SurfacePoint &sp, NormalOut &no)
	{
		S scene(U);
		NormalOutput n;
		n.normal_sample_dist = no.sample_dist;
		n.normal = float3(no.normal);
		n.use_normal = no.use_normal;
		scene.map_normal(geometry_of(sp), n);
		no.sample_dist = n.normal_sample_dist;
		no.normal = n.normal;
		no.use_normal = n.use_normal;
	}
	// map_light, once per lit hit (pshader_sdf.hlsl:505-516)
	static SDF_HD void lights(const FrameU &U, const SurfacePoint &sp, Light *L, bool *used, float &ambient)
	{
		S scene(U);
		LightOutput out[LIGHT_COUNT];
		for (int i = 0; i < LIGHT_COUNT; ++i)
		{
			out[i].used = false;
			out[i].pos = float4(0.f);
			out[i].color = float3(0.f);
			out[i].falloff = 0.f;
			out[i].extend = 0.f;
		}
		scene.map_light(geometry_of(sp), out, ambient);
		for (int i = 0; i < LIGHT_COUNT; ++i)
		{
			used[i] = out[i].used;
			L[i].pos = V3(out[i].pos.x, out[i].pos.y, out[i].pos.z);
			L[i].directional = out[i].pos.w == 1.f;
			L[i].extend = out[i].extend;
			L[i].color = out[i].color;
			L[i].falloff = out[i].falloff;
		}
	}
	static SDF_HD vec3 background(const FrameU &U, vec3 dir, uint32_t iter)
	{
		S scene(U);
		return scene.map_background(float3(dir), iter);
	}
};

} // namespace hlsl
} // namespace sdfr
