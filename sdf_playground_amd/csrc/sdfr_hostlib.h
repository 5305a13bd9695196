// sdfr_hostlib.h -- host-side pieces above the kernels: the FPS camera that produces the
// eye/front/right/top basis, and the shader-variable table with its VAR_ tag parser.
//
// They mirror the reference's host classes so that a user of the reference finds the same
// behaviour behind the C ABI:
//   Camera                 Engine/Camera.{h,cpp} (FPS mode), Engine/SDFRenderer.cpp:85-95
//   ShaderVariableManager  Engine/ShaderUtil.{h,cpp}:49-75,112-277, Engine/ShaderVariable.h
#pragma once
#include <cctype>
#include <cmath>
#include <map>
#include <string>
#include <string_view>
#include <vector>

namespace sdfr {
namespace host {

struct Vec3
{
	float x = 0.f, y = 0.f, z = 0.f;
	Vec3() {}
	Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
	Vec3 operator+(const Vec3 &o) const { return Vec3(x + o.x, y + o.y, z + o.z); }
	Vec3 operator-(const Vec3 &o) const { return Vec3(x - o.x, y - o.y, z - o.z); }
	Vec3 operator*(float s) const { return Vec3(x * s, y * s, z * s); }
	// cross product (the reference spells it operator^)
	Vec3 cross(const Vec3 &o) const { return Vec3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
	Vec3 normalized() const
	{
		float s = 1.f / sqrtf(x * x + y * y + z * z);
		return Vec3(x * s, y * s, z * s);
	}
};

// rotate v about `axis` by `angle` (Rodrigues matrix, row-vector convention of the reference)
inline Vec3 rotate_about_axis(const Vec3 &axis, float angle, const Vec3 &v)
{
	const float c = cosf(angle), s = sinf(angle), ic = 1.f - c;
	const Vec3 n = axis.normalized();
	const float m00 = ic * n.x * n.x + c, m01 = ic * n.y * n.x + s * n.z, m02 = ic * n.z * n.x - s * n.y;
	const float m10 = ic * n.x * n.y - s * n.z, m11 = ic * n.y * n.y + c, m12 = ic * n.z * n.y + s * n.x;
	const float m20 = ic * n.x * n.z + s * n.y, m21 = ic * n.y * n.z - s * n.x, m22 = ic * n.z * n.z + c;
	// homogeneous row/column of a pure rotation: translation 0, w = 1
	const float w = 0.f * v.x + 0.f * v.y + 0.f * v.z + 1.f;
	const float inv_w = 1.f / w;
	return Vec3((m00 * v.x + m01 * v.y + m02 * v.z + 0.f) * inv_w, (m10 * v.x + m11 * v.y + m12 * v.z + 0.f) * inv_w,
		(m20 * v.x + m21 * v.y + m22 * v.z + 0.f) * inv_w);
}

// First-person camera: world up is +y, optional roll about the view direction.
class Camera
{
public:
	void SetEye(const Vec3 &e) { eye = e; }
	const Vec3 &GetEye() const { return eye; }
	void SetLookat(const Vec3 &lookat) { dir = (lookat - eye).normalized(); }
	void SetDirection(const Vec3 &d) { dir = d.normalized(); }
	const Vec3 &GetDirection() const { return dir; }
	void SetAspect(float a) { aspect = a; }
	void SetFOVY(float f) { fovy = f; }
	void SetRoll(float r) { roll = r; }
	Vec3 GetRelXAxis() const { return Vec3(0.f, 1.f, 0.f).cross(dir).normalized(); }
	Vec3 GetRelYAxis() const { return dir.cross(GetRelXAxis()).normalized(); }
	// corners of the view frustum at unit distance: 0 top right, 1 bottom right, 2 bottom left, 3 top left
	Vec3 GetFrustrumEdge(unsigned index) const
	{
		if (index > 3) index = 3;
		static const float flip[] = {1, 1, -1, 1, -1, -1, 1, -1};
		const Vec3 up = rotate_about_axis(dir, roll, GetRelYAxis());
		const Vec3 side = rotate_about_axis(dir, roll, GetRelXAxis());
		return dir + up * tanf(fovy / 2.f) * flip[2 * index] + side * tanf(fovy / 2.f) * aspect * flip[2 * index + 1];
	}
	// what SDFRenderer::render writes into the camera constant buffer
	void GetBasis(Vec3 &out_eye, Vec3 &front, Vec3 &right, Vec3 &top) const
	{
		out_eye = eye;
		front = dir;
		right = (GetFrustrumEdge(0) - GetFrustrumEdge(3)) * 0.5f;
		top = (GetFrustrumEdge(0) - GetFrustrumEdge(1)) * 0.5f;
	}

private:
	Vec3 eye, dir = Vec3(0.f, 0.f, 1.f);
	float roll = 0.f, fovy = 1.04719755f, aspect = 1.5f;
};

// ---- shader variables ------------------------------------------------------------------------
struct Variable
{
	float minval, maxval, start, step;
	float value;
};
using VariableMap = std::map<std::string, Variable, std::less<>>;

inline std::string_view trim_token(std::string_view s)
{
	size_t b = 0;
	while (b < s.size() && isspace((unsigned char)s[b])) ++b;
	size_t e = b;
	while (e < s.size() && !isspace((unsigned char)s[e])) ++e;
	return s.substr(b, e - b);
}

// split `input` at every occurrence of open...close (close may be empty): returns the pieces
// between the separators and the separators themselves
inline void split_tagged(std::string_view input, std::string_view open, std::string_view close, std::vector<std::string_view> &pieces,
	std::vector<std::string_view> &tags)
{
	size_t cur = 0;
	while (true)
	{
		size_t a = input.find(open, cur);
		if (a == std::string_view::npos) break;
		size_t b = input.find(close, a + open.size());
		if (b == std::string_view::npos) break;
		b += close.size();
		pieces.push_back(input.substr(cur, a - cur));
		tags.push_back(input.substr(a, b - a));
		cur = b;
	}
	pieces.push_back(input.substr(cur));
}

class ShaderVariableManager
{
public:
	// Collects every VAR_name(key = value, ...) tag of `text`.  Defaults: min 0, max 2,
	// start = (min+max)/2, step = (max-min)*0.05; unknown keys are ignored; a repeated name
	// keeps its last definition; a tag ends at the first ')'.
	bool parseFile(const std::string &text)
	{
		std::vector<std::string_view> code, tags;
		split_tagged(text, "VAR_", ")", code, tags);
		for (std::string_view tag : tags)
		{
			const size_t lb = tag.find('('), rb = tag.find(')');
			const std::string name(tag.substr(4, lb - 4));
			const std::string_view args = tag.substr(lb + 1, rb - lb - 1);
			std::vector<std::string_view> params, commas;
			split_tagged(args, ",", "", params, commas);
			std::map<std::string, float> kv;
			for (std::string_view p : params)
			{
				std::vector<std::string_view> sides, eqs;
				split_tagged(p, "=", "", sides, eqs);
				if (sides.size() != 2)
				{
					if (eqs.size() == 1) return false;
					break;
				}
				kv[std::string(trim_token(sides[0]))] = std::stof(std::string(trim_token(sides[1])));
			}
			Variable v;
			auto get = [&](const char *k, float dflt) { auto it = kv.find(k); return it != kv.end() ? it->second : dflt; };
			v.minval = get("min", 0.f);
			v.maxval = get("max", 2.f);
			v.start = get("start", (v.maxval + v.minval) * 0.5f);
			v.step = get("step", (v.maxval - v.minval) * 0.05f);
			v.value = v.start;
			variables[name] = v;
		}
		return true;
	}
	bool hasVariables() const { return !variables.empty(); }
	VariableMap &getVariables() { return variables; }
	const VariableMap &getVariables() const { return variables; }
	// unknown names are ignored
	bool setValue(std::string_view name, float val)
	{
		auto it = variables.find(name);
		if (it == variables.end()) return false;
		it->second.value = val;
		return true;
	}
	// the constant-buffer image: values in map order
	std::vector<float> packed() const
	{
		std::vector<float> out;
		for (const auto &kv : variables) out.push_back(kv.second.value);
		return out;
	}

private:
	VariableMap variables;
};

} // namespace host
} // namespace sdfr
