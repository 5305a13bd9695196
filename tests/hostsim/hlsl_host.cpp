// tests/hostsim/hlsl_host.cpp -- TEST-ONLY: a scene in the reference's dialect, translated by the library
// (sdfr_translate_scene_hlsl) and compiled for the CPU with the product's per-pixel pipeline, so that the CPU test tier can
// bit-compare it with the oracle.  Built once per scene with -DSDFR_HLSL_SCENE_FILE="<generated file>" and the scene's VAR_
// macros in -include'd form.  The product never loads this.
#include "sdfr_hostframe.h"
#include "sdfr_hlsl.h"

#include <atomic>
#include <thread>
#include <vector>

namespace sdfr {
#include SDFR_HLSL_SCENE_FILE
} // namespace sdfr

using namespace sdfr;
typedef CachedRayStore<LocalRayStore> HostStore;

extern "C" int hlslsim_render(FrameU *frame, float *out_rgba, unsigned *out_stats, int nthreads)
{
	frame_derive(*frame, -1); // no ahead-of-time scene: only the frame-level derivations (sky rotation, debug plane, extension lights)
	Scene::prepare(*frame);   // a scene in the library's C++ form may stage per-frame constants (the HLSL adapter's is empty)
	const FrameU U = *frame;
	const bool dbg = frame_needs_debug(U);
	std::atomic<int> next_row(0);
	auto worker = [&]() {
		for (;;)
		{
			const int y = next_row.fetch_add(1);
			if (y >= U.height) break;
			for (int x = 0; x < U.width; ++x)
			{
				PixelCounters c = {0, 0, 0};
				LocalRayStore backing;
				HostStore store(backing);
				const vec4 v = dbg ? render_pixel<Scene, true, HostStore>(U, x, y, c, store) : render_pixel<Scene, false, HostStore>(U, x, y, c, store);
				const size_t idx = (size_t)y * U.width + x;
				out_rgba[4 * idx + 0] = v.x;
				out_rgba[4 * idx + 1] = v.y;
				out_rgba[4 * idx + 2] = v.z;
				out_rgba[4 * idx + 3] = v.w;
				if (out_stats)
				{
					out_stats[3 * idx + 0] = c.rays;
					out_stats[3 * idx + 1] = c.march_evals;
					out_stats[3 * idx + 2] = c.hits;
				}
			}
		}
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
	worker();
	for (auto &th : pool) th.join();
	return 0;
}
extern "C" int hlslsim_frame_size() { return (int)sizeof(FrameU); }
