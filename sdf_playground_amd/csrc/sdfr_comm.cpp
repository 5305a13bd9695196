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
// GPU never load it.  If the process already holds an RCCL (PyTorch's), dlopen by SONAME returns
// that one, so a process never runs two.
#include "sdfr_handle.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

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
	bool ok = false;
	std::string why;
};

Rccl &rccl()
{
	static Rccl h;
	static std::once_flag once;
	std::call_once(once, [] {
		const char *names[] = {getenv("SDFR_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char *n : names)
		{
			if (!n || !n[0]) continue;
			if ((h.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
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
};

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
	if (!id_out) return SDFR_ERR_INVALID_ARGUMENT;
	int rc = need_rccl(nullptr);
	if (rc != SDFR_OK) return rc;
	static_assert(sizeof(ncclUniqueId) == SDFR_COMM_ID_BYTES, "SDFR_COMM_ID_BYTES");
	ncclUniqueId id;
	SDFR_NCCL(nullptr, rccl().get_unique_id(&id));
	memcpy(id_out, &id, sizeof id);
	return SDFR_OK;
}

int sdfr_comm_create(const void *id, int rank, int world, int device_ordinal, sdfr_comm **out)
{
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
}

int sdfr_comm_create_all(const int *device_ordinals, int n, sdfr_comm **out_n)
{
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
}

void sdfr_comm_destroy(sdfr_comm *c)
{
	if (!c) return;
	if (c->comm && rccl().ok)
	{
		(void)hipSetDevice(c->device);
		(void)rccl().comm_destroy(c->comm);
	}
	delete c;
}

int sdfr_comm_rank(const sdfr_comm *c) { return c ? c->rank : SDFR_ERR_INVALID_ARGUMENT; }
int sdfr_comm_world(const sdfr_comm *c) { return c ? c->world : SDFR_ERR_INVALID_ARGUMENT; }
const char *sdfr_comm_last_error(const sdfr_comm *c) { return c ? c->error.c_str() : g_comm_error.c_str(); }

int sdfr_comm_selftest(sdfr_comm *c, size_t bytes, void *hip_stream)
{
	if (!c || bytes == 0 || bytes > ((size_t)1 << 30)) return SDFR_ERR_INVALID_ARGUMENT;
	hipStream_t stream = (hipStream_t)hip_stream;
	if (hipSetDevice(c->device) != hipSuccess) return comm_fail(c, "hipSetDevice failed");
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
}

int sdfr_render_gather(sdfr_renderer *r, sdfr_comm *c, int width, int height, void *root_image, int image_format, int wire_format)
{
	GatherShape g;
	int rc = gather_check(r, c, width, height, root_image, image_format, wire_format, g);
	if (rc != SDFR_OK) return rc;
	rc = gather_render(r, c, g);
	if (rc != SDFR_OK) return rc;
	if (g.world > 1 && g.rank_bytes)
	{
		Rccl &n = rccl();
		ncclResult_t nrc = n.group_start();
		if (nrc != ncclSuccess) return fail(r, nccl_fail(c, nrc, "ncclGroupStart"), c->error);
		rc = gather_transfer(r, c, g);
		nrc = n.group_end();
		if (rc != SDFR_OK) return rc;
		if (nrc != ncclSuccess) return fail(r, nccl_fail(c, nrc, "ncclGroupEnd"), c->error);
	}
	return gather_finish(r, c, g, root_image);
}

int sdfr_render_gather_all(sdfr_renderer *const *r, sdfr_comm *const *c, int n, int width, int height, void *root_image, int image_format,
	int wire_format)
{
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
}

} // extern "C"
