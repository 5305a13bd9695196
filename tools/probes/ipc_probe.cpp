// Feasibility probe (developer tool): can two PROCESSES on this box share device memory through hipIpc handles,
// a memcpy from one into the other's buffer, and flag words polled by a kernel?  parent = "root", child = "peer".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d, pid %d)\n", #x, hipGetErrorString(e_), __LINE__, (int)getpid()); _exit(10); } } while (0)

__global__ void k_signal(volatile unsigned *flag, unsigned value)
{
	__threadfence_system();
	__hip_atomic_store((unsigned *)flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait(volatile unsigned *flag, unsigned value, unsigned *timed_out)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__hip_atomic_load((unsigned *)flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value)
	{
		if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { *timed_out = 1; return; } // 2 s
		__builtin_amdgcn_s_sleep(64);
	}
}
__global__ void k_fill(unsigned *p, unsigned n, unsigned seed) { for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i * 2654435761u + seed; }

int main()
{
	int to_child[2], to_parent[2];
	if (pipe(to_child) || pipe(to_parent)) return 2;
	const unsigned n = 4u << 20; // 16 MB
	pid_t pid = fork(); // BEFORE any HIP call in either process
	if (pid == 0)
	{
		hipIpcMemHandle_t hb, hf;
		if (read(to_child[0], &hb, sizeof hb) != (ssize_t)sizeof hb || read(to_child[0], &hf, sizeof hf) != (ssize_t)sizeof hf) _exit(3);
		CK(hipSetDevice(0));
		unsigned *remote = nullptr, *flags = nullptr, *local = nullptr;
		CK(hipIpcOpenMemHandle((void **)&remote, hb, hipIpcMemLazyEnablePeerAccess));
		CK(hipIpcOpenMemHandle((void **)&flags, hf, hipIpcMemLazyEnablePeerAccess));
		CK(hipMalloc((void **)&local, n * 4));
		hipStream_t s;
		CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		for (unsigned frame = 1; frame <= 3; ++frame)
		{
			hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, s, local, n, frame);
			CK(hipMemcpyAsync(remote, local, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
			hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, flags, frame);
			CK(hipStreamSynchronize(s));
			char c = 0;
			if (read(to_child[0], &c, 1) != 1) _exit(4); // root has checked this frame
		}
		CK(hipIpcCloseMemHandle(remote));
		CK(hipIpcCloseMemHandle(flags));
		_exit(0);
	}
	CK(hipSetDevice(0));
	unsigned *buf = nullptr, *flags = nullptr, *timed_out = nullptr;
	CK(hipMalloc((void **)&buf, n * 4));
	CK(hipMalloc((void **)&flags, 256));
	CK(hipMalloc((void **)&timed_out, 4));
	CK(hipMemset(flags, 0, 256));
	CK(hipMemset(timed_out, 0, 4));
	hipIpcMemHandle_t hb, hf;
	CK(hipIpcGetMemHandle(&hb, buf));
	CK(hipIpcGetMemHandle(&hf, flags));
	if (write(to_child[1], &hb, sizeof hb) != (ssize_t)sizeof hb || write(to_child[1], &hf, sizeof hf) != (ssize_t)sizeof hf) return 5;
	unsigned *host = (unsigned *)malloc((size_t)n * 4);
	int bad = 0;
	for (unsigned frame = 1; frame <= 3; ++frame)
	{
		hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, 0, flags, frame, timed_out);
		CK(hipMemcpy(host, buf, (size_t)n * 4, hipMemcpyDeviceToHost)); // after the wait kernel (same stream)
		unsigned to = 0;
		CK(hipMemcpy(&to, timed_out, 4, hipMemcpyDeviceToHost));
		unsigned wrong = 0;
		for (unsigned i = 0; i < n; ++i) wrong += host[i] != i * 2654435761u + frame;
		printf("frame %u: timed_out %u, wrong words %u of %u\n", frame, to, wrong, n);
		bad += (to || wrong) ? 1 : 0;
		char c = 1;
		if (write(to_child[1], &c, 1) != 1) return 6;
	}
	int status = 0;
	waitpid(pid, &status, 0);
	printf("child exit %d\n", WEXITSTATUS(status));
	return bad || WEXITSTATUS(status) ? 1 : 0;
}
