// tests/cpp/host_gather.cpp -- a plain C++ host that shards a frame over every GPU it can see WITHOUT PyTorch:
// one process, one renderer handle and one RCCL communicator per device (sdfr_comm_create_all =
// ncclCommInitAll), sdfr_render_gather_all per frame, the image assembled on device 0; checked against a
// direct render on device 0, bit for bit, for both image formats and with a private-strip split.
// usage: host_gather <width> <height> [max devices]     exit status 0 = identical
#include "sdfr.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(call) \
	do { const int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, r.empty() ? "" : sdfr_last_error(r[0])); return 10; } } while (0)

int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	const int W = atoi(argv[1]), H = atoi(argv[2]);
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { fprintf(stderr, "no device\n"); return 3; }
	if (argc > 3 && atoi(argv[3]) < n) n = atoi(argv[3]);
	std::vector<sdfr_renderer *> r;
	std::vector<int> devices;
	for (int i = 0; i < n; ++i)
	{
		sdfr_renderer *h = nullptr;
		if (sdfr_create(i, &h) != SDFR_OK) return 4;
		r.push_back(h);
		devices.push_back(i);
		CHECK(sdfr_load_scene(h, "labyrinth"));
		sdfr_limits lim;
		CHECK(sdfr_get_limits(h, &lim));
		lim.iter_count = 160;
		CHECK(sdfr_set_limits(h, &lim));
		const float eye[3] = {1.2f, 5.f, 0.4f}, dir[3] = {0.9f, -0.35f, 0.3f};
		CHECK(sdfr_set_camera_direction(h, eye, dir, 1.0471976f, (float)W / (float)H, 0.f));
		CHECK(sdfr_set_time(h, 0.75f));
	}
	std::vector<sdfr_comm *> comm((size_t)n, nullptr);
	if (sdfr_comm_create_all(devices.data(), n, comm.data()) != SDFR_OK) { fprintf(stderr, "comm: %s\n", sdfr_comm_last_error(nullptr)); return 5; }
	// the self-test is a blocking ring exchange: from one thread it can only be run on a world of one
	if (n == 1 && sdfr_comm_selftest(comm[0], 1 << 16, nullptr) != SDFR_OK) { fprintf(stderr, "selftest: %s\n", sdfr_comm_last_error(comm[0])); return 6; }

	int bad = 0;
	const size_t pixels = (size_t)W * H;
	for (int pass = 0; pass < 4; ++pass)
	{
		const int fmt = pass & 1 ? SDFR_RGBA16F : SDFR_RGBA32F, wire = pass & 1 ? SDFR_STRIP_RGB16F_A8 : SDFR_STRIP_RGB32F_A8;
		const size_t bytes = pixels * (fmt == SDFR_RGBA32F ? 16 : 8);
		for (int i = 0; i < n; ++i) CHECK(sdfr_set_strip_split(r[(size_t)i], pass >= 2 ? 5 : 0, 16));
		void *d_direct = nullptr, *d_gathered = nullptr;
		if (hipSetDevice(0) != hipSuccess || hipMalloc(&d_direct, bytes) != hipSuccess || hipMalloc(&d_gathered, bytes) != hipSuccess) return 7;
		(void)hipMemset(d_gathered, 0xee, bytes);
		CHECK(sdfr_render(r[0], W, H, d_direct, fmt, 0, nullptr));
		sdfr_stats whole;
		CHECK(sdfr_get_stats(r[0], &whole));
		for (int rep = 0; rep < 2; ++rep) CHECK(sdfr_render_gather_all(r.data(), comm.data(), n, W, H, d_gathered, fmt, wire));
		unsigned long long rays = 0;
		for (int i = 0; i < n; ++i)
		{
			CHECK(sdfr_sync(r[(size_t)i]));
			sdfr_stats st;
			CHECK(sdfr_get_stats(r[(size_t)i], &st));
			rays += st.rays;
		}
		std::vector<unsigned char> a(bytes), b(bytes);
		(void)hipSetDevice(0);
		if (hipMemcpy(a.data(), d_direct, bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(b.data(), d_gathered, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 8;
		const bool same = memcmp(a.data(), b.data(), bytes) == 0 && rays == whole.rays;
		printf("devices %d format %d wire %d split %d/16: %s (rays %llu vs %llu)\n", n, fmt, wire, pass >= 2 ? 5 : 0, same ? "identical" : "DIFFERENT", rays,
			(unsigned long long)whole.rays);
		bad += same ? 0 : 1;
		(void)hipFree(d_direct);
		(void)hipFree(d_gathered);
	}
	for (int i = 0; i < n; ++i)
	{
		sdfr_comm_destroy(comm[(size_t)i]);
		sdfr_destroy(r[(size_t)i]);
	}
	return bad ? 1 : 0;
}
