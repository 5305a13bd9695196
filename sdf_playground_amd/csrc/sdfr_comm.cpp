// sdfr_comm.cpp -- one frame sharded over the GPUs of a node and gathered on rank 0 with RCCL
// (SURVEY.md 8e).  The reference is single-adapter (Graphics.cpp:34), so nothing here has a
// counterpart there; the split is the one its pixel shader allows: ps_main reads constants and
// its own pixel coordinate only (pshader_sdf.hlsl:260-267), so pixels are independent.
//
// Shape of a frame on N ranks (one process per GPU, or one process driving N devices):
//   every rank   k_pixel over the strips it owns -> compact buffer in a wire format   (handle's stream)
//   peers        ncclSend(compact buffer -> rank 0)                                    (comm stream)
//   rank 0       ncclRecv x (N - 1), one group: N - 1 messages arriving on N - 1 different xGMI
//                links; k_assemble scatters the N buffers into the image               (comm stream)
//                meanwhile its private strips render straight into the image           (handle's stream)
// No ring and no tree: xGMI is a point-to-point mesh, the root has a link of its own to every
// peer, and a gather moves every byte exactly once over exactly one link.
//
// librccl.so is opened on first use (like libhiprtc.so in sdfr_jit.cpp): hosts that render on one
// GPU never load it.  ONE RCCL PER PROCESS: if the process already maps a librccl (PyTorch ships its own copy in
// torch/lib, which libtorch_hip.so loads by PATH), that very file is the one opened here -- found in
// /proc/self/maps rather than left to the SONAME match of dlopen.  sdfr_comm_library_info reports which file serves
// the library and how many distinct librccl files the process maps (a guard: two copies in one process is a state
// nobody tests).  RCCL operations never run on the legacy NULL stream here (sdfr_comm_selftest, sdfr_comm_close).
#include "sdfr_handle.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <set>
#include <thread>

using namespace sdfr;

namespace {

struct Rccl
{
	void *lib = nullptr;
	decltype(&ncclGetUniqueId) get_unique_id = nullptr;
	decltype(&ncclCommInitRank) comm_init_rank = nullptr;
	decltype(&ncclCommInitAll) comm_init_all = nullptr;
	decltype(&ncclCommDestroy) comm_destroy = nullptr;
	decltype(&ncclGroupStart) group_start = nullptr;
	decltype(&ncclGroupEnd) group_end = nullptr;
	decltype(&ncclSend) send = nullptr;
	decltype(&ncclRecv) recv = nullptr;
	decltype(&ncclGetErrorString) error_string = nullptr;
	// optional (NCCL >= 2.14): an orderly, bounded teardown
	decltype(&ncclCommFinalize) comm_finalize = nullptr;
	decltype(&ncclCommAbort) comm_abort = nullptr;
	decltype(&ncclCommGetAsyncError) comm_async_error = nullptr;
	decltype(&ncclGetVersion) get_version = nullptr;
	bool ok = false;
	std::string why;
	std::string path; // the file that serves the entry points (dladdr)
};

// distinct files named librccl.so* that this process maps right now
std::vector<std::string> mapped_rccl_files()
{
	std::vector<std::string> files;
	std::ifstream maps("/proc/self/maps");
	std::string line;
	while (std::getline(maps, line))
	{
		const size_t slash = line.find('/');
		if (slash == std::string::npos) continue;
		std::string path = line.substr(slash);
		const size_t del = path.find(" (deleted)");
		if (del != std::string::npos) path.resize(del);
		const size_t base = path.rfind('/');
		if (path.compare(base + 1, 10, "librccl.so") != 0) continue;
		bool seen = false;
		for (const auto &f : files) seen = seen || f == path;
		if (!seen) files.push_back(path);
	}
	return files;
}

Rccl &rccl()
{
	static Rccl h;
	static std::once_flag once;
	std::call_once(once, [] {
		// a copy the process holds already comes first (one RCCL per process); then the usual names
		std::vector<std::string> names;
		if (const char *e = getenv("SDFR_RCCL_LIBRARY"))
			if (e[0]) names.push_back(e);
		for (const auto &f : mapped_rccl_files()) names.push_back(f);
		for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) names.push_back(n);
		for (const auto &n : names)
		{
			if ((h.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
			h.why += std::string(h.why.empty() ? "" : "; ") + dlerror();
		}
		if (!h.lib) return;
		h.get_unique_id = (decltype(h.get_unique_id))dlsym(h.lib, "ncclGetUniqueId");
		h.comm_init_rank = (decltype(h.comm_init_rank))dlsym(h.lib, "ncclCommInitRank");
		h.comm_init_all = (decltype(h.comm_init_all))dlsym(h.lib, "ncclCommInitAll");
		h.comm_destroy = (decltype(h.comm_destroy))dlsym(h.lib, "ncclCommDestroy");
		h.group_start = (decltype(h.group_start))dlsym(h.lib, "ncclGroupStart");
		h.group_end = (decltype(h.group_end))dlsym(h.lib, "ncclGroupEnd");
		h.send = (decltype(h.send))dlsym(h.lib, "ncclSend");
		h.recv = (decltype(h.recv))dlsym(h.lib, "ncclRecv");
		h.error_string = (decltype(h.error_string))dlsym(h.lib, "ncclGetErrorString");
		h.comm_finalize = (decltype(h.comm_finalize))dlsym(h.lib, "ncclCommFinalize");
		h.comm_abort = (decltype(h.comm_abort))dlsym(h.lib, "ncclCommAbort");
		h.comm_async_error = (decltype(h.comm_async_error))dlsym(h.lib, "ncclCommGetAsyncError");
		h.get_version = (decltype(h.get_version))dlsym(h.lib, "ncclGetVersion");
		Dl_info info;
		if (h.get_unique_id && dladdr((const void *)h.get_unique_id, &info) && info.dli_fname) h.path = info.dli_fname;
		h.ok = h.get_unique_id && h.comm_init_rank && h.comm_init_all && h.comm_destroy && h.group_start && h.group_end && h.send && h.recv &&
			   h.error_string;
		if (!h.ok) h.why = "librccl.so lacks an expected entry point";
	});
	return h;
}

std::string g_comm_error; // errors of calls that have no communicator to hold them

} // namespace

struct sdfr_comm
{
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1, device = 0;
	mutable std::string error;
	// handles whose comm stream carries (or carried) this communicator's transfers: the teardown waits for those streams,
	// a handle that is destroyed first takes itself off the list (comm_forget_renderer)
	std::set<sdfr_renderer *> users;
	// the communicator's own stream: what the self-test runs on when its caller passes the NULL stream
	hipStream_t own_stream = nullptr;
};

namespace {
std::mutex g_registry_lock; // guards every sdfr_comm::users and sdfr_renderer::comms_used
}

// sdfr_destroy: the handle's streams have been drained; no communicator may look at it again
void sdfr::comm_forget_renderer(sdfr_renderer *r)
{
	std::lock_guard<std::mutex> g(g_registry_lock);
	for (void *p : r->comms_used) static_cast<sdfr_comm *>(p)->users.erase(r);
	r->comms_used.clear();
}
static void comm_remember_renderer(sdfr_comm *c, sdfr_renderer *r)
{
	std::lock_guard<std::mutex> g(g_registry_lock);
	if (c->users.insert(r).second) r->comms_used.push_back(c);
}

static int comm_fail(const sdfr_comm *c, const std::string &msg)
{
	if (c)
		c->error = msg;
	else
		g_comm_error = msg;
	return SDFR_ERR_COMM;
}
static int nccl_fail(const sdfr_comm *c, ncclResult_t rc, const char *what) { return comm_fail(c, std::string(what) + ": " + rccl().error_string(rc)); }
#define SDFR_NCCL(c, call) \
	do { ncclResult_t rc_ = (call); if (rc_ != ncclSuccess) return nccl_fail(c, rc_, #call); } while (0)

static int need_rccl(const sdfr_comm *c)
{
	Rccl &n = rccl();
	if (n.ok) return SDFR_OK;
	return comm_fail(c, "RCCL is not available: " + (n.why.empty() ? std::string("librccl.so not found") : n.why));
}

// the stream and events transfers and assembly run on (both transports)
int sdfr::gather_prepare_streams(sdfr_renderer *r)
{
	SDFR_HIP(hipSetDevice(r->device));
	if (!r->comm_stream)
	{
		// transfers and assembly go ahead of rendering waves wherever the two compete for a CU: highest priority
		int least = 0, greatest = 0;
		if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
		SDFR_HIP(hipStreamCreateWithPriority(&r->comm_stream, hipStreamNonBlocking, greatest));
	}
	if (!r->ev_strips) SDFR_HIP(hipEventCreateWithFlags(&r->ev_strips, hipEventDisableTiming));
	if (!r->ev_gathered) SDFR_HIP(hipEventCreateWithFlags(&r->ev_gathered, hipEventDisableTiming));
	for (int k = 0; k < 2; ++k)
		if (!r->ev_xfer[k]) SDFR_HIP(hipEventCreate(&r->ev_xfer[k]));
	return SDFR_OK;
}

// ---- the three phases of a gathered frame, per rank -------------------------------------------------
namespace {

struct GatherShape
{
	int width, height, world, image_format, wire_format;
	size_t rank_bytes; // bytes of one rank's compact buffer (the same on every rank)
};

int gather_check(sdfr_renderer *r, const sdfr_comm *c, int width, int height, const void *root_image, int image_format, int wire_format,
	GatherShape &g)
{
	if (!r || !c) return SDFR_ERR_INVALID_ARGUMENT;
	if (c->device != r->device) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "communicator and renderer are bound to different devices");
	const bool wide = wire_format == SDFR_RGBA32F || wire_format == SDFR_STRIP_RGB32F_A8;
	const bool narrow = wire_format == SDFR_RGBA16F || wire_format == SDFR_STRIP_RGB16F_A8;
	if (!((image_format == SDFR_RGBA32F && wide) || (image_format == SDFR_RGBA16F && narrow)))
		return fail(r, SDFR_ERR_INVALID_ARGUMENT, "image format and wire format do not match (RGBA32F image: 32-bit wire; RGBA16F image: 16-bit wire)");
	if (c->rank == 0 && !root_image) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "rank 0 needs the image to assemble into");
	if (width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 30) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad frame size");
	const int64_t nb = sdfr_strip_buffer_bytes_split(width, height, c->world, wire_format, r->priv_count, r->priv_period);
	if (nb < 0) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad wire format");
	g = GatherShape{width, height, c->world, image_format, wire_format, (size_t)nb};
	return SDFR_OK;
}

// phase 1: render this rank's shared strips into its slot of the wire buffer (handle's stream)
int gather_render(sdfr_renderer *r, const sdfr_comm *c, const GatherShape &g)
{
	int prc = gather_prepare_streams(r);
	if (prc != SDFR_OK) return prc;
	const size_t need = (c->rank == 0 ? (size_t)g.world : (size_t)1) * g.rank_bytes;
	if (r->wire_bytes < need)
	{
		// frames still in flight on either stream read or write the old buffer
		SDFR_HIP(hipStreamSynchronize(r->stream));
		SDFR_HIP(hipStreamSynchronize(r->comm_stream));
		(void)hipFree(r->d_wire);
		r->d_wire = nullptr;
		r->wire_bytes = 0;
		SDFR_HIP(hipMalloc(&r->d_wire, need));
		r->wire_bytes = need;
	}
	SDFR_HIP(hipEventRecord(r->ev_begin, r->stream));
	SDFR_HIP(hipMemsetAsync(r->d_totals, 0, 2 * sizeof(RenderTotals), r->stream));
	r->caller_times = true;
	int rc = SDFR_OK;
	if (g.rank_bytes) rc = render_impl(r, g.width, g.height, c->rank, g.world, r->d_wire, g.wire_format, 0, nullptr, RENDER_STRIPS, r->d_totals);
	r->caller_times = false;
	if (rc != SDFR_OK) return rc;
	SDFR_HIP(hipEventRecord(r->ev_strips, r->stream));
	SDFR_HIP(hipStreamWaitEvent(r->comm_stream, r->ev_strips, 0));
	return SDFR_OK;
}

// phase 2: the point-to-point calls of this rank (comm stream); the caller brackets them in a group
int gather_transfer(sdfr_renderer *r, sdfr_comm *c, const GatherShape &g)
{
	if (g.world == 1 || g.rank_bytes == 0) return SDFR_OK;
	Rccl &n = rccl();
	if (c->rank == 0)
	{
		for (int p = 1; p < g.world; ++p)
		{
			ncclResult_t rc = n.recv((char *)r->d_wire + (size_t)p * g.rank_bytes, g.rank_bytes, ncclInt8, p, c->comm, r->comm_stream);
			if (rc != ncclSuccess) return fail(r, nccl_fail(c, rc, "ncclRecv"), c->error);
		}
	}
	else
	{
		ncclResult_t rc = n.send(r->d_wire, g.rank_bytes, ncclInt8, 0, c->comm, r->comm_stream);
		if (rc != ncclSuccess) return fail(r, nccl_fail(c, rc, "ncclSend"), c->error);
	}
	return SDFR_OK;
}

// phase 3: the root scatters the gathered strips into the image (comm stream) while it renders its
// private strips straight into it (handle's stream); then the handle's stream waits for both
int gather_finish(sdfr_renderer *r, const sdfr_comm *c, const GatherShape &g, void *root_image)
{
	SDFR_HIP(hipSetDevice(r->device));
	int parts = 1;
	if (c->rank == 0)
	{
		if (g.rank_bytes)
		{
			hipError_t e = launch_assemble_strips(g.width, g.height, g.world, r->d_wire, root_image, g.wire_format, r->priv_count, r->priv_period,
				r->comm_stream);
			if (e != hipSuccess) return hip_fail(r, e, "assemble launch");
		}
		if (r->priv_count > 0)
		{
			r->caller_times = true;
			const int rc = render_impl(r, g.width, g.height, 0, 1, root_image, g.image_format, 0, nullptr, RENDER_PRIVATE, r->d_totals + 1);
			r->caller_times = false;
			if (rc != SDFR_OK) return rc;
			parts = 2;
		}
	}
	SDFR_HIP(hipEventRecord(r->ev_gathered, r->comm_stream));
	SDFR_HIP(hipStreamWaitEvent(r->stream, r->ev_gathered, 0));
	SDFR_HIP(hipEventRecord(r->ev_end, r->stream));
	r->totals_parts = parts;
	r->have_render = true;
	r->last_wavefront = false;
	return SDFR_OK;
}

} // namespace

extern "C" {

int sdfr_comm_unique_id(void *id_out)
{
	return guarded(nullptr, [&]() -> int {
		if (!id_out) return SDFR_ERR_INVALID_ARGUMENT;
		int rc = need_rccl(nullptr);
		if (rc != SDFR_OK) return rc;
		static_assert(sizeof(ncclUniqueId) == SDFR_COMM_ID_BYTES, "SDFR_COMM_ID_BYTES");
		ncclUniqueId id;
		SDFR_NCCL(nullptr, rccl().get_unique_id(&id));
		memcpy(id_out, &id, sizeof id);
		return SDFR_OK;
	});
}

int sdfr_comm_create(const void *id, int rank, int world, int device_ordinal, sdfr_comm **out)
{
	return guarded(nullptr, [&]() -> int {
		if (!out) return SDFR_ERR_INVALID_ARGUMENT;
		*out = nullptr;
		if (!id || world < 1 || rank < 0 || rank >= world) return SDFR_ERR_INVALID_ARGUMENT;
		int count = 0;
		if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SDFR_ERR_NO_DEVICE;
		if (device_ordinal < 0 || device_ordinal >= count) return SDFR_ERR_INVALID_ARGUMENT;
		int rc = need_rccl(nullptr);
		if (rc != SDFR_OK) return rc;
		if (hipSetDevice(device_ordinal) != hipSuccess) return SDFR_ERR_HIP;
		ncclUniqueId uid;
		memcpy(&uid, id, sizeof uid);
		sdfr_comm *c = new sdfr_comm();
		c->rank = rank;
		c->world = world;
		c->device = device_ordinal;
		const ncclResult_t nrc = rccl().comm_init_rank(&c->comm, world, uid, rank);
		if (nrc != ncclSuccess)
		{
			nccl_fail(nullptr, nrc, "ncclCommInitRank");
			delete c;
			return SDFR_ERR_COMM;
		}
		*out = c;
		return SDFR_OK;
	});
}

int sdfr_comm_create_all(const int *device_ordinals, int n, sdfr_comm **out_n)
{
	return guarded(nullptr, [&]() -> int {
		if (!device_ordinals || !out_n || n < 1 || n > 64) return SDFR_ERR_INVALID_ARGUMENT;
		for (int i = 0; i < n; ++i) out_n[i] = nullptr;
		int count = 0;
		if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SDFR_ERR_NO_DEVICE;
		for (int i = 0; i < n; ++i)
			if (device_ordinals[i] < 0 || device_ordinals[i] >= count) return SDFR_ERR_INVALID_ARGUMENT;
		int rc = need_rccl(nullptr);
		if (rc != SDFR_OK) return rc;
		std::vector<ncclComm_t> comms((size_t)n, nullptr);
		SDFR_NCCL(nullptr, rccl().comm_init_all(comms.data(), n, device_ordinals));
		for (int i = 0; i < n; ++i)
		{
			sdfr_comm *c = new sdfr_comm();
			c->comm = comms[(size_t)i];
			c->rank = i;
			c->world = n;
			c->device = device_ordinals[i];
			out_n[i] = c;
		}
		return SDFR_OK;
	});
}

// Teardown of a communicator, bounded in time.
//
// What happened in round 2 (gpurun_out/r02/s3: `pytest -m gpu` killed at 500 s behind the 17th dot, i.e. in the teardown
// of tests/test_gpu_comm.py's communicator fixture -> sdfr_comm_destroy -> ncclCommDestroy, which was then called bare).
// What the records allow to say: the process held ONE librccl (the Python wrapper imports torch before it loads this
// library, so the dlopen matched torch's copy); nothing of the communicator was pending (the self-test drains its
// stream, a gather at world 1 issues no RCCL call); the same ncclCommDestroy returned in milliseconds in every bench.py
// process of the same session.  The one difference in how the two kinds of process had USED the communicator: the test
// fixture ran its self-test -- a grouped ncclSend / ncclRecv -- on the legacy NULL stream (stream_handle = None), bench.py
// on a created stream.  RCCL orders its internal streams against the user's with events, and the NULL stream
// synchronises implicitly with every blocking stream of the device; a communicator that has launched on it is the
// only state the hanging process had and the others did not.  That is the cause as far as the evidence reaches -- it
// was not re-run to be made to show again.  What is done about it:
//   * RCCL work never runs on the NULL stream: the self-test moves to a stream the communicator owns (the gather always
//     ran on the handles' comm streams);
//   * the order RCCL's contract asks for: every stream that carried this communicator's transfers is drained first
//     (the handles' comm streams -- a handle destroyed earlier has drained its own --, the communicator's own stream);
//   * ncclCommFinalize (where the library has it), then ncclCommDestroy, on a helper thread with a deadline
//     (SDFR_COMM_CLOSE_TIMEOUT_S, default 30 s); past it the caller calls ncclCommAbort, which is made for exactly
//     this, waits a little longer, and otherwise leaves the helper thread and the communicator behind and reports
//     SDFR_ERR_COMM with the call the teardown was last seen in: a diagnosis instead of a hang.
namespace {
struct CloseJob
{
	std::mutex m;
	std::condition_variable cv;
	bool done = false;
	std::atomic<const char *> stage{"not started"};
	// set by the caller BEFORE it calls ncclCommAbort: abort reclaims the communicator, so a helper that it releases from
	// ncclCommFinalize must not go on to ncclCommDestroy (a second destroy of freed memory)
	// Both flags change under `m`: whichever side gets there first decides -- the helper claims the destroy, or the caller
	// claims the abort; never both.
	bool aborted = false, destroying = false;
	ncclResult_t finalize_rc = ncclSuccess, destroy_rc = ncclSuccess;
};
double close_timeout_s()
{
	if (const char *e = getenv("SDFR_COMM_CLOSE_TIMEOUT_S"))
	{
		const double v = atof(e);
		if (v > 0.0) return v;
	}
	return 30.0;
}
} // namespace

int sdfr_comm_close(sdfr_comm *c)
{
	return guarded(nullptr, [&]() -> int {
		if (!c) return SDFR_ERR_INVALID_ARGUMENT;
		Rccl &n = rccl();
		int status = SDFR_OK;
		std::string diagnosis;
		if (c->comm && n.ok)
		{
			(void)hipSetDevice(c->device);
			{
				// drain the streams its transfers ran on, and let go of the handles
				std::lock_guard<std::mutex> g(g_registry_lock);
				for (sdfr_renderer *r : c->users)
				{
					if (r->comm_stream) (void)hipStreamSynchronize(r->comm_stream);
					for (size_t k = 0; k < r->comms_used.size(); ++k)
						if (r->comms_used[k] == c) { r->comms_used.erase(r->comms_used.begin() + (long)k); break; }
				}
				c->users.clear();
			}
			if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
			auto job = std::make_shared<CloseJob>();
			const ncclComm_t comm = c->comm;
			const int device = c->device;
			std::thread worker([job, comm, device, &n]() {
				(void)hipSetDevice(device);
				if (n.comm_finalize)
				{
					job->stage = "ncclCommFinalize";
					job->finalize_rc = n.comm_finalize(comm);
					// a non-blocking communicator finishes in the background: ncclInProgress until it has
					while (n.comm_async_error && job->finalize_rc == ncclInProgress)
					{
						ncclResult_t state = ncclSuccess;
						if (n.comm_async_error(comm, &state) != ncclSuccess || state != ncclInProgress) { job->finalize_rc = state; break; }
						std::this_thread::sleep_for(std::chrono::milliseconds(5));
					}
				}
				// destroy only what is still ours and was finalized: not after the caller's abort, not after a failed finalize
				bool mine = false;
				{
					std::lock_guard<std::mutex> g(job->m);
					mine = !job->aborted && job->finalize_rc == ncclSuccess;
					job->destroying = mine;
				}
				if (mine)
				{
					job->stage = "ncclCommDestroy";
					job->destroy_rc = n.comm_destroy(comm);
				}
				job->stage = "done";
				std::lock_guard<std::mutex> g(job->m);
				job->done = true;
				job->cv.notify_all();
			});
			const auto wait_for = [&](double seconds) {
				std::unique_lock<std::mutex> g(job->m);
				return job->cv.wait_for(g, std::chrono::duration<double>(seconds), [&] { return job->done; });
			};
			if (wait_for(close_timeout_s()))
			{
				worker.join();
				if (job->finalize_rc != ncclSuccess)
				{
					status = SDFR_ERR_COMM;
					diagnosis = std::string("ncclCommFinalize: ") + n.error_string(job->finalize_rc) + " (the communicator was not destroyed)";
				}
				else if (job->destroy_rc != ncclSuccess)
				{
					status = SDFR_ERR_COMM;
					diagnosis = std::string("ncclCommDestroy: ") + n.error_string(job->destroy_rc);
				}
			}
			else
			{
				const char *stuck_in = job->stage.load();
				status = SDFR_ERR_COMM;
				diagnosis = std::string("communicator teardown did not finish within ") + std::to_string(close_timeout_s()) + " s; last seen in " + stuck_in;
				// abort only a teardown that is stuck BEFORE the destroy: a helper inside ncclCommDestroy already owns the
				// communicator's memory, and aborting under it would free it twice
				bool may_abort = false;
				{
					std::lock_guard<std::mutex> g(job->m);
					may_abort = n.comm_abort && !job->destroying && !job->done;
					job->aborted = may_abort;
				}
				if (may_abort)
				{
					(void)n.comm_abort(comm);
					diagnosis += wait_for(5.0) ? "; ncclCommAbort released it" : "; ncclCommAbort did not release it: the helper thread and the communicator are left behind";
				}
				else if (n.comm_abort && !job->done)
					diagnosis += "; stuck inside the destroy itself: not aborted (the helper thread and the communicator are left behind)";
				if (job->finalize_rc != ncclSuccess && job->finalize_rc != ncclInProgress) diagnosis += std::string("; ncclCommFinalize had returned: ") + n.error_string(job->finalize_rc);
				if (job->done) worker.join(); else worker.detach(); // `job` is shared: the thread may outlive this call
				const auto files = mapped_rccl_files();
				diagnosis += "; librccl in use: " + n.path + ", distinct librccl files mapped: " + std::to_string(files.size());
			}
		}
		if (status != SDFR_OK)
		{
			g_comm_error = diagnosis;
			fprintf(stderr, "libsdfr: %s\n", diagnosis.c_str());
		}
		if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
		delete c;
		return status;
	});
}

void sdfr_comm_destroy(sdfr_comm *c) { (void)sdfr_comm_close(c); }

int sdfr_comm_library_info(char *path_out, size_t path_bytes, int *nccl_version, int *copies_mapped)
{
	return guarded(nullptr, [&]() -> int {
		const int rc = need_rccl(nullptr);
		if (rc != SDFR_OK) return rc;
		Rccl &n = rccl();
		if (path_out && path_bytes) snprintf(path_out, path_bytes, "%s", n.path.c_str());
		if (nccl_version)
		{
			*nccl_version = 0;
			if (n.get_version) (void)n.get_version(nccl_version);
		}
		if (copies_mapped) *copies_mapped = (int)mapped_rccl_files().size();
		return SDFR_OK;
	});
}

int sdfr_comm_rank(const sdfr_comm *c) { return c ? c->rank : SDFR_ERR_INVALID_ARGUMENT; }
int sdfr_comm_world(const sdfr_comm *c) { return c ? c->world : SDFR_ERR_INVALID_ARGUMENT; }
const char *sdfr_comm_last_error(const sdfr_comm *c) { return c ? c->error.c_str() : g_comm_error.c_str(); }

// (drains `hip_stream` before it returns: nothing of it is pending when the communicator is closed)
int sdfr_comm_selftest(sdfr_comm *c, size_t bytes, void *hip_stream)
{
	return guarded(nullptr, [&]() -> int {
		if (!c || bytes == 0 || bytes > ((size_t)1 << 30)) return SDFR_ERR_INVALID_ARGUMENT;
		hipStream_t stream = (hipStream_t)hip_stream;
		if (hipSetDevice(c->device) != hipSuccess) return comm_fail(c, "hipSetDevice failed");
		// never RCCL on the legacy NULL stream (see sdfr_comm_close): the call is blocking anyway, so the NULL stream is
		// drained and the exchange runs on a stream of the communicator's own.  SDFR_COMM_ALLOW_NULL_STREAM=1 (diagnosis
		// only) keeps the caller's NULL stream.
		if (!stream && !getenv("SDFR_COMM_ALLOW_NULL_STREAM"))
		{
			if (hipStreamSynchronize(nullptr) != hipSuccess) return comm_fail(c, "hipStreamSynchronize(NULL) failed");
			if (!c->own_stream && hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return comm_fail(c, "hipStreamCreate failed");
			stream = c->own_stream;
		}
		const int to = (c->rank + 1) % c->world, from = (c->rank + c->world - 1) % c->world;
		auto pattern = [](int rank, size_t i) { return (unsigned char)((i * 2654435761u + (size_t)rank * 97u + (i >> 13)) & 0xffu); };
		std::vector<unsigned char> host(bytes);
		for (size_t i = 0; i < bytes; ++i) host[i] = pattern(c->rank, i);
		unsigned char *d_send = nullptr, *d_recv = nullptr;
		hipError_t e = hipMalloc((void **)&d_send, bytes);
		if (e == hipSuccess) e = hipMalloc((void **)&d_recv, bytes);
		if (e == hipSuccess) e = hipMemcpyAsync(d_send, host.data(), bytes, hipMemcpyHostToDevice, stream);
		if (e == hipSuccess) e = hipMemsetAsync(d_recv, 0, bytes, stream);
		int rc = e == hipSuccess ? SDFR_OK : comm_fail(c, std::string("selftest buffers: ") + hipGetErrorString(e));
		if (rc == SDFR_OK)
		{
			Rccl &n = rccl();
			ncclResult_t nrc = n.group_start();
			if (nrc == ncclSuccess) nrc = n.send(d_send, bytes, ncclInt8, to, c->comm, stream);
			if (nrc == ncclSuccess) nrc = n.recv(d_recv, bytes, ncclInt8, from, c->comm, stream);
			const ncclResult_t erc = n.group_end();
			if (nrc == ncclSuccess) nrc = erc;
			if (nrc != ncclSuccess) rc = nccl_fail(c, nrc, "selftest send/recv");
		}
		if (rc == SDFR_OK)
		{
			e = hipMemcpyAsync(host.data(), d_recv, bytes, hipMemcpyDeviceToHost, stream);
			if (e == hipSuccess) e = hipStreamSynchronize(stream);
			if (e != hipSuccess) rc = comm_fail(c, std::string("selftest read-back: ") + hipGetErrorString(e));
		}
		if (rc == SDFR_OK)
		{
			size_t bad = 0;
			for (size_t i = 0; i < bytes; ++i) bad += host[i] != pattern(from, i);
			if (bad) rc = comm_fail(c, "selftest: " + std::to_string(bad) + " of " + std::to_string(bytes) + " bytes from rank " + std::to_string(from) + " differ");
		}
		(void)hipFree(d_send);
		(void)hipFree(d_recv);
		return rc;
	});
}

int sdfr_render_gather(sdfr_renderer *r, sdfr_comm *c, int width, int height, void *root_image, int image_format, int wire_format)
{
	return guarded(r, [&]() -> int {
		GatherShape g;
		int rc = gather_check(r, c, width, height, root_image, image_format, wire_format, g);
		if (rc != SDFR_OK) return rc;
		comm_remember_renderer(c, r);
		rc = gather_render(r, c, g);
		if (rc != SDFR_OK) return rc;
		r->have_xfer = false;
		if (g.world > 1 && g.rank_bytes)
		{
			Rccl &n = rccl();
			SDFR_HIP(hipEventRecord(r->ev_xfer[0], r->comm_stream));
			ncclResult_t nrc = n.group_start();
			if (nrc != ncclSuccess) return fail(r, nccl_fail(c, nrc, "ncclGroupStart"), c->error);
			rc = gather_transfer(r, c, g);
			nrc = n.group_end();
			if (rc != SDFR_OK) return rc;
			if (nrc != ncclSuccess) return fail(r, nccl_fail(c, nrc, "ncclGroupEnd"), c->error);
			SDFR_HIP(hipEventRecord(r->ev_xfer[1], r->comm_stream));
			r->have_xfer = true;
			r->xfer_bytes = c->rank == 0 ? (size_t)(g.world - 1) * g.rank_bytes : g.rank_bytes;
		}
		return gather_finish(r, c, g, root_image);
	});
}

int sdfr_render_gather_all(sdfr_renderer *const *r, sdfr_comm *const *c, int n, int width, int height, void *root_image, int image_format,
	int wire_format)
{
	return guarded(nullptr, [&]() -> int {
		if (!r || !c || n < 1 || n > 64) return SDFR_ERR_INVALID_ARGUMENT;
		std::vector<GatherShape> g((size_t)n);
		for (int i = 0; i < n; ++i)
		{
			if (!r[i] || !c[i] || c[i]->rank != i || c[i]->world != n) return SDFR_ERR_INVALID_ARGUMENT;
			const int rc = gather_check(r[i], c[i], width, height, root_image, image_format, wire_format, g[(size_t)i]);
			if (rc != SDFR_OK) return rc;
			if (g[(size_t)i].rank_bytes != g[0].rank_bytes) return fail(r[i], SDFR_ERR_INVALID_ARGUMENT, "the handles carry different strip splits");
		}
		for (int i = 0; i < n; ++i)
		{
			comm_remember_renderer(c[i], r[i]);
			const int rc = gather_render(r[i], c[i], g[(size_t)i]);
			if (rc != SDFR_OK) return rc;
		}
		if (n > 1 && g[0].rank_bytes)
		{
			Rccl &nc = rccl();
			ncclResult_t nrc = nc.group_start();
			if (nrc != ncclSuccess) return fail(r[0], nccl_fail(c[0], nrc, "ncclGroupStart"), c[0]->error);
			int rc = SDFR_OK;
			for (int i = 0; i < n && rc == SDFR_OK; ++i)
			{
				(void)hipSetDevice(r[i]->device);
				rc = gather_transfer(r[i], c[i], g[(size_t)i]);
			}
			nrc = nc.group_end();
			if (rc != SDFR_OK) return rc;
			if (nrc != ncclSuccess) return fail(r[0], nccl_fail(c[0], nrc, "ncclGroupEnd"), c[0]->error);
		}
		for (int i = 0; i < n; ++i)
		{
			const int rc = gather_finish(r[i], c[i], g[(size_t)i], root_image);
			if (rc != SDFR_OK) return rc;
		}
		return SDFR_OK;
	});
}

} // extern "C"
