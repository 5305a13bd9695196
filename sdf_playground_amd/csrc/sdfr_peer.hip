// sdfr_peer.hip -- the gather of a sharded frame WITHOUT a collective library: peers copy their strips straight
// into rank 0's buffer through hipIpc mappings (option iii in its copy form).
//
//   rank 0   owns the region: the gathered buffer (world slots) and a few flag words, both exported with
//            hipIpcGetMemHandle; renders its own strips into slot 0
//   peers    map the region; per frame: render strips -> local buffer, wait until rank 0 has released the
//            region (it assembled the frame before), hipMemcpyAsync(local -> slot[rank]) -- a DMA over the
//            link, no compute unit involved --, then a one-thread kernel stores the frame number into
//            arrived[rank] with system scope
//   rank 0   a one-wave kernel waits until every arrived[p] has reached the frame number, k_assemble scatters
//            the slots into the image, a one-thread kernel stores the frame number into `released`
// Waits are bounded (2 s of the 100-MHz clock): a rank that never arrives makes the wait give up instead of hanging
// the GPU.  A wait that gave up says so in a status word of this rank's own -- twice: in host memory that the GPU writes
// through a mapping (the host reads it without a synchronisation) and in device memory, which is what the kernels that
// follow on the stream look at first (every thread of the assembly reads it: over PCIe that was 17 ms per 4K frame) --
// and those are then skipped: the copy into rank 0's slot and rank 0's
// assembly are SKIPPED (no torn frame: a peer must not write a slot rank 0 may still be reading), the next
// sdfr_render_gather_peer refuses with SDFR_ERR_COMM, sdfr_peer_region_status reports it, and only a new region
// (create / open) clears it.  The flag words other devices write while a kernel polls them are fine-grained memory.
// Same strip layout, wire formats, private strips and counters as sdfr_render_gather (sdfr_comm.cpp); the reference
// has no counterpart (one adapter, Graphics.cpp:34).
#include "sdfr_handle.h"

#include <cstring>

using namespace sdfr;

namespace {

enum { FLAG_ARRIVED = 0, FLAG_RELEASED = 64, FLAG_WORDS = 128 };
#define PEER_WAIT_TICKS 200000000ull // 2 s

struct Descriptor
{
	hipIpcMemHandle_t buffer, flags; // 64 bytes each
	uint64_t capacity;
	int32_t world, device;
	uint32_t magic, reserved;
};
static_assert(sizeof(Descriptor) <= SDFR_PEER_REGION_BYTES, "SDFR_PEER_REGION_BYTES");

// lanes 0 .. n-1 each wait for flags[first + lane] >= value; every wave of the grid reaches the end.  A wait that gives
// up (or finds that an earlier one has) leaves the frame number in *status: what follows on the stream is skipped.
__global__ void k_peer_wait(uint32_t *flags, int first, int n, uint32_t value, uint32_t *status, uint32_t *gave_up, uint32_t frame)
{
	const int lane = (int)threadIdx.x;
	if (lane >= n) return;
	if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__hip_atomic_load(flags + first + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value)
	{
		if (__builtin_amdgcn_s_memrealtime() - t0 > PEER_WAIT_TICKS)
		{
			__hip_atomic_store(status, frame ? frame : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(gave_up, frame ? frame : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		__builtin_amdgcn_s_sleep(32);
	}
}

// a peer's strips into its slot of rank 0's buffer, over the link -- unless a wait gave up (see the head of the file)
__global__ __launch_bounds__(256) void k_peer_copy(uint32_t *dst, const uint32_t *src, size_t words, const uint32_t *status)
{
	if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
	const size_t stride = (size_t)gridDim.x * blockDim.x, quads = words / 4;
	const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
	uint4 *d4 = reinterpret_cast<uint4 *>(dst);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride) d4[i] = s4[i];
	for (size_t i = quads * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) dst[i] = src[i];
}
__global__ void k_peer_signal_unless(uint32_t *flag, uint32_t value, const uint32_t *status)
{
	if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
	__threadfence_system();
	__hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void region_forget(sdfr_renderer *r)
{
	r->peer_buffer = nullptr;
	r->peer_flags = nullptr;
	r->peer_capacity = 0;
	r->peer_world = 0;
	r->peer_owner = false;
	r->peer_frame = 0;
}

// this rank's status word: host memory the GPU writes through a mapping, so the host reads it without a synchronisation
int status_word(sdfr_renderer *r)
{
	if (!r->peer_status)
	{
		SDFR_HIP(hipHostMalloc((void **)&r->peer_status, 64, hipHostMallocMapped));
		SDFR_HIP(hipHostGetDevicePointer((void **)&r->d_peer_status, r->peer_status, 0));
	}
	if (!r->d_peer_gave_up) SDFR_HIP(hipMalloc((void **)&r->d_peer_gave_up, 64));
	*r->peer_status = 0u; // a new region starts clean
	SDFR_HIP(hipMemset(r->d_peer_gave_up, 0, 64));
	return SDFR_OK;
}

// the flag words are written by other devices while a kernel of this one polls them: fine-grained where the runtime has it
hipError_t flags_alloc(uint32_t **flags, bool *fine)
{
	*fine = true;
	hipError_t e = hipExtMallocWithFlags((void **)flags, FLAG_WORDS * sizeof(uint32_t), hipDeviceMallocFinegrained);
	if (e != hipSuccess)
	{
		(void)hipGetLastError();
		*fine = false;
		e = hipMalloc((void **)flags, FLAG_WORDS * sizeof(uint32_t));
	}
	return e;
}

} // namespace

extern "C" {

int sdfr_peer_region_close(sdfr_renderer *r)
{
	if (!r) return SDFR_ERR_INVALID_ARGUMENT;
	if (!r->peer_buffer) return SDFR_OK;
	(void)hipSetDevice(r->device);
	(void)hipStreamSynchronize(r->stream);
	if (r->comm_stream) (void)hipStreamSynchronize(r->comm_stream);
	if (r->peer_owner)
	{
		(void)hipFree(r->peer_buffer);
		(void)hipFree(r->peer_flags);
	}
	else
	{
		(void)hipIpcCloseMemHandle(r->peer_buffer);
		(void)hipIpcCloseMemHandle(r->peer_flags);
	}
	region_forget(r);
	return SDFR_OK;
}

int sdfr_peer_region_create(sdfr_renderer *r, size_t capacity_bytes, int world, void *descriptor_out)
{
	if (!r || !descriptor_out || capacity_bytes == 0 || world < 1 || world > 64) return SDFR_ERR_INVALID_ARGUMENT;
	(void)sdfr_peer_region_close(r);
	SDFR_HIP(hipSetDevice(r->device));
	int src = status_word(r);
	if (src != SDFR_OK) return src;
	void *buffer = nullptr;
	uint32_t *flags = nullptr;
	SDFR_HIP(hipMalloc(&buffer, capacity_bytes));
	bool fine = false;
	hipError_t e = flags_alloc(&flags, &fine);
	if (e == hipSuccess) e = hipMemset(flags, 0, FLAG_WORDS * sizeof(uint32_t));
	Descriptor d;
	memset(&d, 0, sizeof d);
	if (e == hipSuccess) e = hipIpcGetMemHandle(&d.buffer, buffer);
	if (e == hipSuccess) e = hipIpcGetMemHandle(&d.flags, flags);
	if (e != hipSuccess && fine && flags)
	{
		// a runtime that cannot export fine-grained memory: ordinary device memory, as before
		(void)hipGetLastError();
		(void)hipFree(flags);
		flags = nullptr;
		fine = false;
		e = hipMalloc((void **)&flags, FLAG_WORDS * sizeof(uint32_t));
		if (e == hipSuccess) e = hipMemset(flags, 0, FLAG_WORDS * sizeof(uint32_t));
		if (e == hipSuccess) e = hipIpcGetMemHandle(&d.buffer, buffer);
		if (e == hipSuccess) e = hipIpcGetMemHandle(&d.flags, flags);
	}
	d.reserved = fine ? 1u : 0u;
	if (e != hipSuccess)
	{
		(void)hipFree(buffer);
		(void)hipFree(flags);
		return hip_fail(r, e, "peer region (hipMalloc / hipIpcGetMemHandle)");
	}
	d.capacity = capacity_bytes;
	d.world = world;
	d.device = r->device;
	d.magic = 0x53444652u;
	memset(descriptor_out, 0, SDFR_PEER_REGION_BYTES);
	memcpy(descriptor_out, &d, sizeof d);
	r->peer_buffer = buffer;
	r->peer_flags = flags;
	r->peer_capacity = capacity_bytes;
	r->peer_world = world;
	r->peer_owner = true;
	r->peer_frame = 0;
	return SDFR_OK;
}

int sdfr_peer_region_open(sdfr_renderer *r, const void *descriptor)
{
	if (!r || !descriptor) return SDFR_ERR_INVALID_ARGUMENT;
	Descriptor d;
	memcpy(&d, descriptor, sizeof d);
	if (d.magic != 0x53444652u || d.world < 1 || d.world > 64 || d.capacity == 0) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "not a peer-region descriptor");
	(void)sdfr_peer_region_close(r);
	SDFR_HIP(hipSetDevice(r->device));
	int src = status_word(r);
	if (src != SDFR_OK) return src;
	void *buffer = nullptr, *flags = nullptr;
	SDFR_HIP(hipIpcOpenMemHandle(&buffer, d.buffer, hipIpcMemLazyEnablePeerAccess));
	hipError_t e = hipIpcOpenMemHandle(&flags, d.flags, hipIpcMemLazyEnablePeerAccess);
	if (e != hipSuccess)
	{
		(void)hipIpcCloseMemHandle(buffer);
		return hip_fail(r, e, "hipIpcOpenMemHandle");
	}
	r->peer_buffer = buffer;
	r->peer_flags = (uint32_t *)flags;
	r->peer_capacity = (size_t)d.capacity;
	r->peer_world = d.world;
	r->peer_owner = false;
	r->peer_frame = 0;
	return SDFR_OK;
}

int sdfr_peer_region_status(sdfr_renderer *r)
{
	if (!r) return SDFR_ERR_INVALID_ARGUMENT;
	if (!r->peer_flags) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "no peer region");
	SDFR_HIP(hipSetDevice(r->device));
	SDFR_HIP(hipStreamSynchronize(r->stream));
	if (r->comm_stream) SDFR_HIP(hipStreamSynchronize(r->comm_stream));
	const uint32_t gave_up = r->peer_status ? *(volatile uint32_t *)r->peer_status : 0u;
	if (gave_up)
		return fail(r, SDFR_ERR_COMM, "a wait of the peer-copy gather gave up in frame " + std::to_string(gave_up) +
									  ": some rank did not arrive (or did not release the region) within 2 s; that frame's copy and assembly were skipped");
	return SDFR_OK;
}

int sdfr_render_gather_peer(sdfr_renderer *r, int rank, int world, int width, int height, void *root_image, int image_format, int wire_format)
{
	if (!r) return SDFR_ERR_INVALID_ARGUMENT;
	if (!r->peer_buffer) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "no peer region: sdfr_peer_region_create (rank 0) / sdfr_peer_region_open (peers) first");
	if (world != r->peer_world || rank < 0 || rank >= world || (rank == 0) != r->peer_owner)
		return fail(r, SDFR_ERR_INVALID_ARGUMENT, "rank / world do not match the peer region (rank 0 owns it)");
	const bool wide = wire_format == SDFR_RGBA32F || wire_format == SDFR_STRIP_RGB32F_A8;
	const bool narrow = wire_format == SDFR_RGBA16F || wire_format == SDFR_STRIP_RGB16F_A8;
	if (!((image_format == SDFR_RGBA32F && wide) || (image_format == SDFR_RGBA16F && narrow)))
		return fail(r, SDFR_ERR_INVALID_ARGUMENT, "image format and wire format do not match (RGBA32F image: 32-bit wire; RGBA16F image: 16-bit wire)");
	if (rank == 0 && !root_image) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "rank 0 needs the image to assemble into");
	if (width < 1 || height < 1 || (int64_t)width * height > (int64_t)1 << 30) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad frame size");
	const int64_t nb64 = sdfr_strip_buffer_bytes_split(width, height, world, wire_format, r->priv_count, r->priv_period);
	if (nb64 < 0) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "bad wire format");
	const size_t nb = (size_t)nb64;
	if (nb * (size_t)world > r->peer_capacity) return fail(r, SDFR_ERR_INVALID_ARGUMENT, "the peer region is too small for this frame");

	// a wait that gave up earlier (no synchronisation: the word is host memory): no further frame through this region
	if (r->peer_status && *(volatile uint32_t *)r->peer_status)
		return fail(r, SDFR_ERR_COMM, "the peer region is stale: a wait gave up in frame " + std::to_string(*(volatile uint32_t *)r->peer_status) +
									  " (a rank did not arrive within 2 s); close it and make a new one");
	int rc = gather_prepare_streams(r);
	if (rc != SDFR_OK) return rc;
	const uint32_t frame = ++r->peer_frame;
	uint32_t *flags = r->peer_flags;
	void *mine = r->peer_buffer; // rank 0 renders straight into slot 0
	if (rank != 0)
	{
		if (r->wire_bytes < nb)
		{
			SDFR_HIP(hipStreamSynchronize(r->stream));
			SDFR_HIP(hipStreamSynchronize(r->comm_stream));
			(void)hipFree(r->d_wire);
			r->d_wire = nullptr;
			r->wire_bytes = 0;
			SDFR_HIP(hipMalloc(&r->d_wire, nb));
			r->wire_bytes = nb;
		}
		mine = r->d_wire;
	}
	SDFR_HIP(hipEventRecord(r->ev_begin, r->stream));
	SDFR_HIP(hipMemsetAsync(r->d_totals, 0, 2 * sizeof(RenderTotals), r->stream));
	// (rank 0: slot 0 is free again -- the handle's stream already waits, since the end of the call before, for the
	// assembly that read it)
	r->caller_times = true;
	rc = nb ? render_impl(r, width, height, rank, world, mine, wire_format, 0, nullptr, RENDER_STRIPS, r->d_totals) : SDFR_OK;
	r->caller_times = false;
	if (rc != SDFR_OK) return rc;
	SDFR_HIP(hipEventRecord(r->ev_strips, r->stream));
	SDFR_HIP(hipStreamWaitEvent(r->comm_stream, r->ev_strips, 0));

	int parts = 1;
	if (rank != 0)
	{
		if (nb && world > 1)
		{
			// the region is free once rank 0 has assembled the frame before out of it
			hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, r->comm_stream, flags, (int)FLAG_RELEASED, 1, frame - 1u, r->d_peer_status, r->d_peer_gave_up, frame);
			// (a kernel, not hipMemcpyAsync: a copy engine cannot be told to skip a frame whose wait gave up)
			hipLaunchKernelGGL(k_peer_copy, dim3(256), dim3(256), 0, r->comm_stream, reinterpret_cast<uint32_t *>((char *)r->peer_buffer + (size_t)rank * nb),
				reinterpret_cast<const uint32_t *>(r->d_wire), nb / 4, r->d_peer_gave_up);
			hipLaunchKernelGGL(k_peer_signal_unless, dim3(1), dim3(1), 0, r->comm_stream, flags + FLAG_ARRIVED + rank, frame, r->d_peer_gave_up);
		}
	}
	else
	{
		if (nb)
		{
			if (world > 1) hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, r->comm_stream, flags, (int)FLAG_ARRIVED + 1, world - 1, frame, r->d_peer_status, r->d_peer_gave_up, frame);
			hipError_t e = launch_assemble_strips(width, height, world, r->peer_buffer, root_image, wire_format, r->priv_count, r->priv_period, r->comm_stream,
				r->d_peer_gave_up);
			if (e != hipSuccess) return hip_fail(r, e, "assemble launch");
		}
		if (world > 1) hipLaunchKernelGGL(k_peer_signal_unless, dim3(1), dim3(1), 0, r->comm_stream, flags + FLAG_RELEASED, frame, r->d_peer_gave_up);
		if (r->priv_count > 0)
		{
			r->caller_times = true;
			rc = render_impl(r, width, height, 0, 1, root_image, image_format, 0, nullptr, RENDER_PRIVATE, r->d_totals + 1);
			r->caller_times = false;
			if (rc != SDFR_OK) return rc;
			parts = 2;
		}
	}
	SDFR_HIP(hipGetLastError());
	SDFR_HIP(hipEventRecord(r->ev_gathered, r->comm_stream));
	SDFR_HIP(hipStreamWaitEvent(r->stream, r->ev_gathered, 0));
	SDFR_HIP(hipEventRecord(r->ev_end, r->stream));
	r->totals_parts = parts;
	r->have_render = true;
	r->last_wavefront = false;
	return SDFR_OK;
}

} // extern "C"
