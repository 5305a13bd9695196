// sdfr_hlsl.cpp -- a scene file in the reference's dialect (Engine/shader/scenes/*.hlsl: map / map_normal / map_light /
// map_background, the OBJECT / OBJECT_TRANSPARENT / MATERIAL macros, pshader_sdf.hlsl:79-84) becomes the body of a C++
// class that sdfr_hlsl.h completes; the run-time compiler (sdfr_jit.cpp) does the rest.  The text is NOT parsed: HLSL's
// expression and statement syntax is C's, and what differs is handled by types (swizzles, vector arithmetic: sdfr_hlsl.h) or
// by the few textual rules below.
#include "sdfr_hlsl_translate.h"

#include <cctype>
#include <regex>

namespace sdfr {

namespace {

// unsuffixed floating literals are floats in HLSL (`0.65`, `1e30`, `5.`): give them the suffix; leaves identifiers,
// integers, suffixed literals, comments and strings alone
std::string float_literals(const std::string &s)
{
	std::string out;
	out.reserve(s.size() + s.size() / 16);
	size_t i = 0;
	const size_t n = s.size();
	auto ident_char = [](char c) { return isalnum((unsigned char)c) || c == '_'; };
	while (i < n)
	{
		const char c = s[i];
		if (c == '/' && i + 1 < n && s[i + 1] == '/') // line comment
		{
			const size_t e = s.find('\n', i);
			out.append(s, i, (e == std::string::npos ? n : e) - i);
			i = e == std::string::npos ? n : e;
			continue;
		}
		if (c == '/' && i + 1 < n && s[i + 1] == '*') // block comment
		{
			size_t e = s.find("*/", i + 2);
			e = e == std::string::npos ? n : e + 2;
			out.append(s, i, e - i);
			i = e;
			continue;
		}
		if (c == '"') // string (an #include's, if any survived)
		{
			size_t e = s.find('"', i + 1);
			e = e == std::string::npos ? n : e + 1;
			out.append(s, i, e - i);
			i = e;
			continue;
		}
		if (isalpha((unsigned char)c) || c == '_') // identifier or keyword, digits included
		{
			size_t e = i;
			while (e < n && ident_char(s[e])) ++e;
			out.append(s, i, e - i);
			i = e;
			continue;
		}
		const bool starts_number = isdigit((unsigned char)c) || (c == '.' && i + 1 < n && isdigit((unsigned char)s[i + 1]) && (out.empty() || !(ident_char(out.back()) || out.back() == ')' || out.back() == ']')));
		if (!starts_number)
		{
			out.push_back(c);
			++i;
			continue;
		}
		size_t e = i;
		bool is_float = false, hex = false;
		if (c == '0' && e + 1 < n && (s[e + 1] == 'x' || s[e + 1] == 'X'))
		{
			hex = true;
			e += 2;
			while (e < n && isxdigit((unsigned char)s[e])) ++e;
		}
		else
		{
			while (e < n && isdigit((unsigned char)s[e])) ++e;
			if (e < n && s[e] == '.' && !(e + 1 < n && (isalpha((unsigned char)s[e + 1]) || s[e + 1] == '_') && s[e + 1] != 'f' && s[e + 1] != 'F' && s[e + 1] != 'h' && s[e + 1] != 'H' && s[e + 1] != 'e' && s[e + 1] != 'E'))
			{
				is_float = true;
				++e;
				while (e < n && isdigit((unsigned char)s[e])) ++e;
			}
			if (e < n && (s[e] == 'e' || s[e] == 'E'))
			{
				size_t x = e + 1;
				if (x < n && (s[x] == '+' || s[x] == '-')) ++x;
				if (x < n && isdigit((unsigned char)s[x]))
				{
					is_float = true;
					e = x;
					while (e < n && isdigit((unsigned char)s[e])) ++e;
				}
			}
		}
		out.append(s, i, e - i);
		if (!hex && is_float)
		{
			if (e < n && (s[e] == 'f' || s[e] == 'F' || s[e] == 'h' || s[e] == 'H' || s[e] == 'l' || s[e] == 'L'))
			{
				out.push_back('f'); // half / double suffixes: a float here
				++e;
			}
			else
				out.push_back('f');
		}
		else
			while (e < n && (s[e] == 'u' || s[e] == 'U' || s[e] == 'l' || s[e] == 'L')) out.push_back(s[e++]);
		i = e;
	}
	return out;
}

} // namespace

std::string hlsl_scene_class_body(const std::string &hlsl)
{
	std::string t = hlsl;
	// the libraries are this library's own (sdfr_hlsl_lib.inl): their #include lines go
	t = std::regex_replace(t, std::regex(R"((^|\n)[ \t]*#[ \t]*include[ \t]+"[^"\n]*"[^\n]*)"), "$1");
	// [unroll], [loop], [branch], [flatten(...)] ...
	t = std::regex_replace(t, std::regex(R"(\[\s*(unroll|loop|branch|flatten|fastopt|allow_uav_condition|call|forcecase)\s*(\(\s*[0-9]*\s*\))?\s*\])"), "");
	// parameters: `inout T name` / `out T name` are references (an array parameter is one already), `in` is the default
	t = std::regex_replace(t, std::regex(R"(\b(?:inout|out)\s+((?:const\s+)?[A-Za-z_]\w*)\s+([A-Za-z_]\w*)\s*\[)"), "$1 $2[");
	t = std::regex_replace(t, std::regex(R"(\b(?:inout|out)\s+((?:const\s+)?[A-Za-z_]\w*)\s+([A-Za-z_]\w*))"), "$1 &$2");
	t = std::regex_replace(t, std::regex(R"(([(,]\s*)in\s+(?=(?:const\s+)?[A-Za-z_]\w*\s+[A-Za-z_]\w*\s*[,)\[]))"), "$1");
	// `static const` globals become members with initialisers (a class body cannot hold non-constant static data, and device
	// code no dynamic initialisation); `static` locals become locals
	t = std::regex_replace(t, std::regex(R"(\bstatic\s+const\b)"), "const");
	t = std::regex_replace(t, std::regex(R"(\bstatic\s+(?=(?:float|int|uint|bool|half)[1-4]?(?:x[1-4])?\b))"), "");
	// D3D's float -> int casts saturate: `(int)(expr)` / `(uint)(expr)`
	t = std::regex_replace(t, std::regex(R"(\(\s*int\s*\)\s*\()"), "ftoi_(");
	t = std::regex_replace(t, std::regex(R"(\(\s*uint\s*\)\s*\()"), "ftou_(");
	// ... and in front of a bare operand: `(int)cell.x`, `(uint)index` (a name with its members; not a call, not an element)
	t = std::regex_replace(t, std::regex(R"(\(\s*int\s*\)\s*([A-Za-z_]\w*(?:\.\w+)*)(?![\w.(\[]))"), "ftoi_($1)");
	t = std::regex_replace(t, std::regex(R"(\(\s*uint\s*\)\s*([A-Za-z_]\w*(?:\.\w+)*)(?![\w.(\[]))"), "ftou_($1)");
	t = std::regex_replace(t, std::regex(R"(\bhalf([1-4]?)\b)"), "float$1");
	return float_literals(t);
}

std::string hlsl_scene_source(const std::string &hlsl)
{
	std::string s;
	s += "// generated by sdfr_hlsl.cpp from a scene in the reference's dialect\n";
	s += "namespace hlsl {\n";
	// the scene's own functions carry no __host__ __device__: everything in the class gets both
	s += "#if defined(__HIP__) || defined(__HIPCC_RTC__)\n#pragma clang force_cuda_host_device begin\n#endif\n";
	s += "struct UserScene\n{\n";
	s += "\tSDFR_HLSL_FRAME_MEMBERS(UserScene)\n";
	s += "#include \"sdfr_hlsl_lib.inl\"\n";
	s += "#line 1 \"scene.hlsl\"\n";
	s += hlsl_scene_class_body(hlsl);
	s += "\n};\n";
	s += "#if defined(__HIP__) || defined(__HIPCC_RTC__)\n#pragma clang force_cuda_host_device end\n#endif\n";
	s += "} // namespace hlsl\n";
	s += "typedef hlsl::SceneAdapter<hlsl::UserScene> Scene;\n";
	return s;
}

} // namespace sdfr
