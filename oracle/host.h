// oracle/host.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restates the host-side producers of the hot path's inputs:
//   * camera basis: Engine/SDFRenderer.cpp:85-95, Engine/Camera.cpp:24-49,156-166,
//     Engine/Math3D.cpp:216-225 (Vector3 * and /), :286-289 (cross), :297-302
//     (Normalized), :769-795 (RotationAxisMatrix), :936-943 (Matrix4x4 * Vector3)
//   * shader variables: Engine/ShaderUtil.cpp:122-191 (parseFile), Engine/Util.cpp:4-49
//     (removeSpaces, splitString), Engine/ShaderVariable.h:6-12
// Pinned by: oracle/_ref (the reference's own Math3D.cpp + Camera.cpp compiled here)
// and by the reference's TestStringSplit1..7 (UnitTest/UnitTest.cpp:91-179).
#pragma once
#include <cmath>
#include <map>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

namespace orc {
namespace host {

struct V3 { float x, y, z; };

inline V3 v_add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 v_sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 v_scale(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
// Math3D.cpp:286-289
inline V3 v_cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// Math3D.cpp:297-302
inline V3 v_normalized(V3 a)
{
	float s = 1.f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
	return V3{a.x * s, a.y * s, a.z * s};
}

// Math3D.cpp:769-795 followed by :936-943 (w == 1 for a pure rotation)
inline V3 rotate_axis(V3 axis, float angle, V3 v)
{
	float c = cosf(angle);
	float s = sinf(angle);
	float inv_c = 1.f - c;
	V3 n = v_normalized(axis);
	float f11 = inv_c * n.x * n.x + c;
	float f21 = inv_c * n.x * n.y - s * n.z;
	float f31 = inv_c * n.x * n.z + s * n.y;
	float f12 = inv_c * n.y * n.x + s * n.z;
	float f22 = inv_c * n.y * n.y + c;
	float f32 = inv_c * n.y * n.z - s * n.x;
	float f13 = inv_c * n.z * n.x - s * n.y;
	float f23 = inv_c * n.z * n.y + s * n.x;
	float f33 = inv_c * n.z * n.z + c;
	float f14 = 0.f, f24 = 0.f, f34 = 0.f, f44 = 1.f;
	float w = f14 * v.x + f24 * v.y + f34 * v.z + f44;
	V3 r{f11 * v.x + f12 * v.y + f13 * v.z + f14, f21 * v.x + f22 * v.y + f23 * v.z + f24, f31 * v.x + f32 * v.y + f33 * v.z + f34};
	float inv_w = 1.f / w;
	return V3{r.x * inv_w, r.y * inv_w, r.z * inv_w};
}

struct CameraBasis { V3 eye, front, right, top; };

// FPS-mode camera (Camera.cpp:36-49) looking along `dir` (normalised here like SetDirection :28-31)
inline CameraBasis camera_from_direction(V3 eye, V3 direction, float fovy, float aspect, float roll)
{
	V3 dir = v_normalized(direction);
	V3 xaxis = v_normalized(v_cross(V3{0.f, 1.f, 0.f}, dir)); // GetRelXAxis
	V3 yaxis = v_normalized(v_cross(dir, xaxis));             // GetRelYAxis
	// GetFrustrumEdge (:156-166)
	const float flip[] = {1, 1, -1, 1, -1, -1, 1, -1};
	V3 edge[4];
	for (unsigned index = 0; index < 4; ++index)
	{
		V3 ry = rotate_axis(dir, roll, yaxis);
		V3 rx = rotate_axis(dir, roll, xaxis);
		V3 a = v_scale(v_scale(ry, tanf(fovy / 2.f)), flip[2 * index]);
		V3 b = v_scale(v_scale(v_scale(rx, tanf(fovy / 2.f)), aspect), flip[2 * index + 1]);
		edge[index] = v_add(v_add(dir, a), b);
	}
	CameraBasis cb;
	cb.eye = eye;
	cb.front = dir;
	// SDFRenderer.cpp:89-90
	cb.right = v_scale(v_sub(edge[0], edge[3]), 0.5f);
	cb.top = v_scale(v_sub(edge[0], edge[1]), 0.5f);
	return cb;
}

// Camera.cpp:24-27
inline CameraBasis camera_lookat(V3 eye, V3 lookat, float fovy, float aspect, float roll)
{
	return camera_from_direction(eye, v_sub(lookat, eye), fovy, aspect, roll);
}

// ---- Util.cpp:4-15 -----------------------------------------------------------------
inline std::string_view removeSpaces(std::string_view input)
{
	size_t i = 0;
	while (i < input.size() && isspace(static_cast<unsigned char>(input[i])))
		++i;
	size_t j = i;
	while (j < input.size() && !isspace(static_cast<unsigned char>(input[j])))
		++j;
	return input.substr(i, j - i);
}

// ---- Util.cpp:17-49 ----------------------------------------------------------------
inline std::pair<std::vector<std::string_view>, std::vector<std::string_view>> splitString(std::string_view input,
	std::string_view pattern_start, std::string_view pattern_end = {})
{
	std::vector<std::string_view> parts, separators;
	std::string_view::size_type current = 0, npos = std::string_view::npos;
	for (;;)
	{
		auto index_start = input.find(pattern_start, current);
		if (index_start == npos)
			break;
		auto index_end = input.find(pattern_end, index_start + pattern_start.size());
		if (index_end == npos)
			break;
		index_end += pattern_end.size();
		parts.push_back(input.substr(current, index_start - current));
		separators.push_back(input.substr(index_start, index_end - index_start));
		current = index_end;
	}
	parts.push_back(input.substr(current));
	return {parts, separators};
}

// ShaderVariable.h:6-12
struct Variable { float minval, maxval, start, step, value; };
typedef std::map<std::string, Variable, std::less<>> VariableMap;

// ShaderUtil.cpp:122-191, CollectPass behaviour: every VAR_name(...) tag is recorded
// (a name seen twice keeps the last definition), unknown keys are ignored, defaults
// min 0, max 2, start (min+max)/2, step (max-min)*0.05 (F10).
inline bool parseVariables(const std::string &input, VariableMap &variables)
{
	const std::string_view var_tag = "VAR_";
	auto [code_blocks, variable_blocks] = splitString(input, var_tag, ")");
	for (const auto &block : variable_blocks)
	{
		auto bracket_begin = block.find("(");
		auto bracket_end = block.find(")");
		auto var_name_short = block.substr(var_tag.size(), bracket_begin - var_tag.size());
		auto param_string = block.substr(bracket_begin + 1, bracket_end - bracket_begin - 1);

		auto [params, unused] = splitString(param_string, ",");
		std::map<std::string, float> param_map;
		for (const auto &param : params)
		{
			auto [parts, separators] = splitString(param, "=");
			if (parts.size() != 2)
			{
				if (separators.size() == 1)
					return false;
				break;
			}
			auto name_str = removeSpaces(parts[0]);
			auto val_str = removeSpaces(parts[1]);
			param_map[std::string(name_str)] = std::stof(std::string(val_str));
		}

		Variable var;
		auto iter = param_map.find("min");
		var.minval = iter != param_map.end() ? iter->second : 0.f;
		iter = param_map.find("max");
		var.maxval = iter != param_map.end() ? iter->second : 2.f;
		iter = param_map.find("start");
		var.start = iter != param_map.end() ? iter->second : (var.maxval + var.minval) * 0.5f;
		iter = param_map.find("step");
		var.step = iter != param_map.end() ? iter->second : (var.maxval - var.minval) * 0.05f;
		var.value = var.start;
		variables[std::string(var_name_short)] = var;
	}
	return true;
}

} // namespace host
} // namespace orc
