// tests/cpp/host_reload.cpp -- the edit-and-reload workflow from C++ (include/sdfr.hpp):
// a scene given as a source file is compiled at run time, rendered, then a broken edit is
// rejected with the compiler's message while the previous scene keeps rendering.
// A file named *.hlsl is a scene in the reference's own dialect (initShaderHlsl), anything else the library's C++ form.
// usage: host_reload <scene source file> <stime> <size> <out.raw> eye(3) target(3)
#include "sdfr.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

int main(int argc, char **argv)
{
	if (argc < 11) return 2;
	std::ifstream in(argv[1]);
	std::stringstream text;
	text << in.rdbuf();
	const int size = atoi(argv[3]);
	sdfr::SDFRenderer renderer;
	if (!renderer.init(0)) return 3;
	const size_t name_len = strlen(argv[1]);
	const bool hlsl = name_len > 5 && strcmp(argv[1] + name_len - 5, ".hlsl") == 0;
	if (!(hlsl ? renderer.initShaderHlsl("from_file", text.str()) : renderer.initShaderSource("from_file", text.str())))
	{
		fprintf(stderr, "%s\n", renderer.lastError());
		return 4;
	}
	for (const auto &kv : renderer.getVariableMap()) printf("%s=%g\n", kv.first.c_str(), kv.second.value);
	sdfr::Camera camera;
	camera.SetEye(sdfr::Vector3((float)atof(argv[5]), (float)atof(argv[6]), (float)atof(argv[7])));
	camera.SetLookat(sdfr::Vector3((float)atof(argv[8]), (float)atof(argv[9]), (float)atof(argv[10])));
	camera.SetAspect(1.f);
	renderer.setParameters((float)atof(argv[2]));
	std::vector<float> img((size_t)size * size * 4), again(img.size());
	if (!renderer.render(camera, size, size, img.data(), SDFR_RGBA32F, true)) { fprintf(stderr, "%s\n", renderer.lastError()); return 5; }
	// a broken edit: the message names the error, the loaded scene stays
	if (hlsl ? renderer.initShaderHlsl("broken", text.str() + "\nthis is not HLSL;\n") : renderer.initShaderSource("broken", text.str() + "\nthis is not C++;\n")) return 6;
	if (!strstr(renderer.lastError(), "error")) return 7;
	if (!renderer.render(camera, size, size, again.data(), SDFR_RGBA32F, true)) return 8;
	if (memcmp(img.data(), again.data(), img.size() * sizeof(float)) != 0) return 9;
	FILE *f = fopen(argv[4], "wb");
	if (!f) return 10;
	fwrite(img.data(), sizeof(float), img.size(), f);
	fclose(f);
	return 0;
}
