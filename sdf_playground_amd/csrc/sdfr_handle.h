// sdfr_handle.h -- the renderer handle behind the C ABI (include/sdfr.h) and the helpers the
// translation units that implement it share (sdfr_api.cpp: single-GPU entries; sdfr_comm.cpp:
// the multi-GPU gather over RCCL).  Host code only.
#pragma once
#include "../../include/sdfr.h"

#include "sdfr_hostframe.h"
#include "sdfr_hostlib.h"
#include "sdfr_jit.h"
#include "sdfr_kernels.h"

#include <cstdio>
#include <exception>
#include <string>
#include <vector>

struct sdfr_renderer
{
	int device = 0;
	hipStream_t stream = nullptr;
	int scene = -1; // index of an ahead-of-time scene, or SDFR_SCENE_COUNT: `jit` holds a scene compiled at run time
	sdfr::JitScene jit;
	int schedule = SDFR_SCHEDULE_PIXEL; // the faster one on every measured scene
	bool profiling = false;
	int tile_w_log2 = 0; // 0: the scene's own tile shape (SceneTileShape, 8 x 8 unless it says otherwise); SDFR_TILE_W_LOG2 sets 3 .. 6
	int launch_mode = 0; // sdfr_set_launch_mode
	int priv_count = 0, priv_period = 1; // sdfr_set_strip_split
	sdfr::FrameU U;
	sdfr::host::ShaderVariableManager vars;
	std::vector<std::string> scene_var_slots; // slot k of FrameU::scene_var <- this variable
	mutable std::string error;

	sdfr::RenderTotals *d_totals = nullptr; // [2]: counters of the last launch; [1] = the private strips of a gather (sdfr_comm.cpp)
	int totals_parts = 1;             // how many of them the last render filled
	sdfr::WavefrontWorkspace ws = {};
	size_t wavefront_capacity = 0; // pixels the wavefront-only part of `ws` is allocated for
	void *d_stage = nullptr; // staging image for host-destination renders
	size_t stage_bytes = 0;
	uint32_t *d_pstat = nullptr;
	size_t pstat_bytes = 0;
	// sdfr_register_host_target: the caller's persistent host image, page-locked with the runtime
	void *pinned_host = nullptr;
	size_t pinned_bytes = 0;

	hipEvent_t ev_begin = nullptr, ev_end = nullptr;
	hipEvent_t ev_post[3] = {}; // before / between / after the two post-processing kernels
	bool have_post = false;
	bool step_shortcuts = true; // sdfr_set_step_shortcuts
	unsigned char *d_post_flags = nullptr; // per row segment: did the horizontal bloom pass store any light (sdfr_post.hip)
	size_t post_flag_bytes = 0;
	double ms_setup = 0.0;      // host time of the last latch_frame (+ Scene::prepare of a run-time scene)
	hipEvent_t ev_march[32] = {}, ev_shade[32] = {};
	int last_rounds = 0;
	bool have_render = false;
	bool last_wavefront = false;
	bool last_profiled = false;

	// multi-GPU gather (sdfr_comm.cpp): send / receive / assembly run on a stream of their own so
	// that the root's private strips render while the peers' strips travel
	hipStream_t comm_stream = nullptr;
	hipEvent_t ev_strips = nullptr, ev_gathered = nullptr;
	hipEvent_t ev_xfer[2] = {nullptr, nullptr}; // around the last gather's transfer on the comm stream ("gather transfer", sdfr_get_timings)
	bool have_xfer = false;
	size_t xfer_bytes = 0;                      // bytes this rank sent (peers) or received (rank 0) in that transfer
	void *d_wire = nullptr;    // this rank's compact strips; on the root: world x that, slot 0 = its own
	size_t wire_bytes = 0;
	bool caller_times = false; // render_impl leaves ev_begin / ev_end to its caller
	std::vector<void *> comms_used; // sdfr_comm* whose transfers ran on comm_stream (sdfr_comm.cpp keeps both sides of the list)

	// Two frames in flight inside one handle (sdfr_set_frames_in_flight).  What a frame in flight owns -- its stream, its
	// workspace (ray queue, counter records, tile cursors with the row order learned from ITS last frame), its counters
	// and its two events -- is a Lane; the members above (`stream`, `ws`, `wavefront_capacity`, `d_totals`, `totals_parts`,
	// `ev_begin`, `ev_end`, `have_render`) are the CURRENT lane's, `other` is the lane of the frame before.  sdfr_render swaps the two
	// before it launches, so everything else in the library keeps working on "the handle's stream and workspace".
	struct Lane
	{
		hipStream_t stream = nullptr;
		sdfr::WavefrontWorkspace ws = {};
		size_t wavefront_capacity = 0;
		sdfr::RenderTotals *d_totals = nullptr;
		int totals_parts = 1;
		hipEvent_t ev_begin = nullptr, ev_end = nullptr;
		bool have_render = false;
		const char *out_lo = nullptr, *out_hi = nullptr; // device range its last frame was rendered into
	};
	int frames_in_flight = 1;
	Lane other;
	hipStream_t lane_streams[2] = {nullptr, nullptr}; // the library's own streams while frames_in_flight == 2
	hipStream_t user_stream = nullptr;                // what sdfr_set_stream gave (in use while frames_in_flight == 1)
	const char *out_lo = nullptr, *out_hi = nullptr;  // the current lane's
};

static inline int fail(const sdfr_renderer *r, int code, const std::string &msg)
{
	if (r) r->error = msg;
	return code;
}
// No C++ exception crosses the C boundary: every extern "C" entry point with a body of more than a line runs inside this
// (std::bad_alloc, std::regex_error from the scene translation, std::system_error from a thread that cannot start ...
// would otherwise reach a C or ctypes caller as std::terminate -> abort()).
template <class F>
static inline int guarded(const sdfr_renderer *r, F body) noexcept
{
	try
	{
		return body();
	}
	catch (const std::exception &e)
	{
		try
		{
			if (r) r->error = std::string("internal error: ") + e.what();
			fprintf(stderr, "libsdfr: internal error: %s\n", e.what());
		}
		catch (...)
		{
		}
		return SDFR_ERR_INTERNAL;
	}
	catch (...)
	{
		return SDFR_ERR_INTERNAL;
	}
}
static inline int hip_fail(const sdfr_renderer *r, hipError_t e, const char *what)
{
	return fail(r, SDFR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define SDFR_HIP(call) \
	do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(r, e_, #call); } while (0)

namespace sdfr {

enum RenderMode { RENDER_FULL, RENDER_STRIPS, RENDER_PRIVATE };

// bytes of a compact image of `pixels` pixels (the packed strip formats are padded to 4)
size_t image_bytes(size_t pixels, int format);
bool is_wire_format(int format);

// one launch of the handle's scene over the rows `mode` selects (sdfr_api.cpp).  `totals` receives
// the launch's counters (nullptr: the handle's d_totals).
int render_impl(sdfr_renderer *r, int width, int height, int rank, int world, void *out, int format, int out_on_host, uint32_t *pixel_stats,
	RenderMode mode, RenderTotals *totals = nullptr);

// the stream and events of a gathered frame (sdfr_comm.cpp)
int gather_prepare_streams(sdfr_renderer *r);
// the handle is going away: communicators that remember it must forget it (sdfr_comm.cpp)
void comm_forget_renderer(sdfr_renderer *r);

} // namespace sdfr
