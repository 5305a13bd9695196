// tests/cpp/host_mirror.cpp -- a C++ user of the reference-shaped host interface
// (include/sdfr.hpp) over the C ABI: renders one golden-fixture frame and writes raw floats.
// usage: host_mirror <scene> <stime> <size> <out.raw> eye(3) target(3) is_dir [var=value ...]
#include "sdfr.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv)
{
	if (argc < 12) return 2;
	const char *scene = argv[1];
	const float stime = (float)atof(argv[2]);
	const int size = atoi(argv[3]);
	sdfr::SDFRenderer renderer;
	if (!renderer.init(0)) { fprintf(stderr, "init failed\n"); return 3; }
	sdfr::Camera camera;
	std::vector<float> img((size_t)size * size * 4);
	// no scene yet: render must report false (SDFRenderer.cpp:70-73)
	if (renderer.render(camera, size, size, img.data(), SDFR_RGBA32F, true)) return 4;
	if (renderer.initShader("no_such_scene")) return 5;
	if (!renderer.initShader(scene)) { fprintf(stderr, "%s\n", renderer.lastError()); return 6; }
	camera.SetEye(sdfr::Vector3((float)atof(argv[5]), (float)atof(argv[6]), (float)atof(argv[7])));
	sdfr::Vector3 t((float)atof(argv[8]), (float)atof(argv[9]), (float)atof(argv[10]));
	if (atoi(argv[11])) camera.SetDirection(t); else camera.SetLookat(t);
	camera.SetAspect(1.f);
	for (int i = 12; i < argc; ++i)
	{
		char *eq = strchr(argv[i], '=');
		if (!eq) continue;
		*eq = 0;
		auto &vars = renderer.getVariableMap();
		auto it = vars.find(std::string_view(argv[i]));
		if (it == vars.end()) return 7;
		it->second.value = (float)atof(eq + 1); // edited in place, like the reference's UI
	}
	renderer.setParameters(stime);
	if (!renderer.render(camera, size, size, img.data(), SDFR_RGBA32F, true)) { fprintf(stderr, "%s\n", renderer.lastError()); return 8; }
	FILE *f = fopen(argv[4], "wb");
	if (!f) return 9;
	fwrite(img.data(), sizeof(float), img.size(), f);
	fclose(f);
	printf("variables:");
	for (const auto &kv : renderer.getVariableMap()) printf(" %s=%g", kv.first.c_str(), kv.second.value);
	printf("\n");
	return 0;
}
