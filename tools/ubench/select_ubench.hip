// select_ubench.hip -- developer micro-benchmark: what does a select cost on gfx950?  v_cndmask in its two encodings
// (VOP2: the mask is VCC, implicitly; VOP3: any SGPR pair or VCC, explicitly), alone and behind the compare that
// forms the mask, at 1/2/4/8 waves per SIMD.  Cycles per wave64 instruction per SIMD at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/select_ubench.hip -o /tmp/select_ubench && /tmp/select_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, ASM)                                                                  \
	__global__ __launch_bounds__(256) void k_##NAME(float *out, int iters, float b, float c) \
	{                                                                                      \
		float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
		for (int i = 0; i < iters; ++i)                                                     \
		{                                                                                  \
			ASM                                                                            \
		}                                                                                  \
		out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;        \
	}

#define CND_E32_VCC(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b));
#define CND_E64_VCC(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b));
#define CND_E64_SGPR(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a##i) : "v"(b));
#define CMP_CND_VCC(i) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a##i) : "v"(b), "v"(c) : "vcc");
#define CMP_CND_VCC64(i) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %2, vcc" : "+v"(a##i) : "v"(b), "v"(c) : "vcc");
#define CMP_CND_SGPR(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(a##i) : "v"(b), "v"(c) : "s20", "s21");
#define CMP_ADD_CND_VCC(i) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_add_f32 %3, %3, %2\n v_add_f32 %3, %3, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a##i) : "v"(b), "v"(c), "v"(a7) : "vcc");
#define CLASS32_CND(i) asm volatile("v_cmp_class_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a##i) : "v"(m), "v"(c) : "vcc");
#define MIN_F32(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define ADD_F32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));

KERNEL(cnd_e32_vcc, REP8(CND_E32_VCC))
KERNEL(cnd_e64_vcc, REP8(CND_E64_VCC))
KERNEL(cnd_e64_sgpr, REP8(CND_E64_SGPR))
KERNEL(cmp_cnd_vcc, REP8(CMP_CND_VCC))
KERNEL(cmp_cnd_vcc64, REP8(CMP_CND_VCC64))
KERNEL(cmp_cnd_sgpr, REP8(CMP_CND_SGPR))
KERNEL(min_f32, REP8(MIN_F32))
KERNEL(add_f32, REP8(ADD_F32))
__global__ __launch_bounds__(256) void k_class32_cnd(float *out, int iters, float b, float c)
{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	int m = 0x264;
	for (int i = 0; i < iters; ++i)
	{
		REP8(CLASS32_CND)
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b;
}
// four compares into four SGPR pairs, then the four selects: the mask is three instructions old when it is read
__global__ __launch_bounds__(256) void k_cmp4_cnd4_sgpr(float *out, int iters, float b, float c)
{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
	for (int i = 0; i < iters; ++i)
	{
		asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[26:27], %3, %4\n"
					 "v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_cndmask_b32_e64 %1, %1, %5, s[22:23]\n v_cndmask_b32_e64 %2, %2, %5, s[24:25]\n v_cndmask_b32_e64 %3, %3, %5, s[26:27]\n"
					 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
// the same with VCC for one of them and a gap of three instructions
__global__ __launch_bounds__(256) void k_cmp_gap3_cnd_vcc(float *out, int iters, float b, float c)
{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
	for (int i = 0; i < iters; ++i)
	{
		asm volatile("v_cmp_lt_f32_e32 vcc, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5\n"
					 "v_cndmask_b32_e32 %0, %0, %5, vcc\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5\n"
					 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void k_cmp_gap3_cnd_sgpr(float *out, int iters, float b, float c)
{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
	for (int i = 0; i < iters; ++i)
	{
		asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5\n"
					 "v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5\n"
					 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s20", "s21");
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

// one compare, then n selects on the same mask (what `cond ? vecA : vecB` compiles to)
#define RUN_KERNEL(NAME, CMP, CND, ...)                                                    \
	__global__ __launch_bounds__(256) void k_##NAME(float *out, int iters, float b, float c) \
	{                                                                                      \
		float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                    \
		for (int i = 0; i < iters; ++i)                                                     \
		{                                                                                  \
			asm volatile(CMP "\n" CND(0) "\n" CND(1) "\n" CND(2) "\n" CND(3) "\n" CMP "\n" CND(1) "\n" CND(0)     \
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : __VA_ARGS__);   \
		}                                                                                  \
		out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;                            \
	}
#define R_E32(k) "v_cndmask_b32_e32 %" #k ", %" #k ", %5, vcc"
#define R_E64V(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %5, vcc"
#define R_E64S(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %5, s[20:21]"
RUN_KERNEL(run_e32, "v_cmp_lt_f32_e32 vcc, %0, %4", R_E32, "vcc")
RUN_KERNEL(run_e64v, "v_cmp_lt_f32_e32 vcc, %0, %4", R_E64V, "vcc")
RUN_KERNEL(run_e64s, "v_cmp_lt_f32_e64 s[20:21], %0, %4", R_E64S, "s20", "s21")

// a transcendental and its first consumer: back to back, or with independent work in between
#define TRANS_KERNEL(NAME, BODY, NCLOB)                                                    \
	__global__ __launch_bounds__(256) void k_##NAME(float *out, int iters, float b, float c) \
	{                                                                                      \
		float a0 = threadIdx.x + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, t0 = 0.f, t1 = 0.f; \
		for (int i = 0; i < iters; ++i)                                                     \
		{                                                                                  \
			asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(t0), "+v"(t1) : "v"(b), "v"(c)); \
		}                                                                                  \
		out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + t0 + t1;                  \
	}
// 8 instructions each: rcp, dependent fma, dependent fma, 5 independent adds -- in two orders
TRANS_KERNEL(rcp_use_now, "v_rcp_f32 %4, %0\n v_fma_f32 %5, -%0, %4, 1.0\n v_fma_f32 %0, %5, %4, %4\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7\n v_add_f32 %3, %3, %7\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7", 0)
TRANS_KERNEL(rcp_use_later, "v_rcp_f32 %4, %0\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7\n v_add_f32 %3, %3, %7\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7\n v_fma_f32 %5, -%0, %4, 1.0\n v_fma_f32 %0, %5, %4, %4", 0)
TRANS_KERNEL(rsq_use_now, "v_rsq_f32 %4, %0\n v_mul_f32 %5, %0, %4\n v_mul_f32 %0, %5, %4\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7\n v_add_f32 %3, %3, %7\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7", 0)
TRANS_KERNEL(rsq_use_later, "v_rsq_f32 %4, %0\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7\n v_add_f32 %3, %3, %7\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7\n v_mul_f32 %5, %0, %4\n v_mul_f32 %0, %5, %4", 0)
TRANS_KERNEL(floor_use_now, "v_floor_f32 %4, %0\n v_mul_f32 %5, %0, %4\n v_mul_f32 %0, %5, %4\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7\n v_add_f32 %3, %3, %7\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7", 0)

typedef void (*kfn)(float *, int, float, float);
struct Entry { const char *name; kfn fn; int per_iter; };

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	printf("device %s, %d CUs\n", prop.name, cus);
	float *out;
	hipMalloc(&out, sizeof(float) * 256 * cus * 8);
	std::vector<Entry> entries = {
		{"add_f32", k_add_f32, 8}, {"min_f32", k_min_f32, 8},
		{"cnd_e32_vcc", k_cnd_e32_vcc, 8}, {"cnd_e64_vcc", k_cnd_e64_vcc, 8}, {"cnd_e64_sgpr", k_cnd_e64_sgpr, 8},
		{"cmp+cnd e32 vcc", k_cmp_cnd_vcc, 16}, {"cmp+cnd e64 vcc", k_cmp_cnd_vcc64, 16}, {"cmp+cnd e64 sgpr", k_cmp_cnd_sgpr, 16},
		{"class32+cnd vcc", k_class32_cnd, 16}, {"4cmp,4cnd sgpr", k_cmp4_cnd4_sgpr, 8},
		{"cmp,3add,cnd,3add vcc", k_cmp_gap3_cnd_vcc, 8}, {"cmp,3add,cnd,3add sgpr", k_cmp_gap3_cnd_sgpr, 8},
		{"rcp,use,use,5add", k_rcp_use_now, 8}, {"rcp,5add,use,use", k_rcp_use_later, 8}, {"rsq,use,use,5add", k_rsq_use_now, 8}, {"rsq,5add,use,use", k_rsq_use_later, 8},
		{"floor,use,use,5add", k_floor_use_now, 8},
		{"cmp,4cnd,cmp,2cnd e32 vcc", k_run_e32, 8}, {"cmp,4cnd,cmp,2cnd e64 vcc", k_run_e64v, 8}, {"cmp,4cnd,cmp,2cnd e64 sgpr", k_run_e64s, 8},
	};
	const int iters = 40000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	printf("%-26s %8s %8s %8s %8s   (cycles per wave64 instruction per SIMD at 2.4 GHz; waves/SIMD = 1,2,4,8)\n", "sequence", "w1", "w2", "w4", "w8");
	for (auto &en : entries)
	{
		printf("%-26s", en.name);
		for (int wps : {1, 2, 4, 8})
		{
			const int blocks = cus * wps;
			hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
			hipDeviceSynchronize();
			hipEventRecord(e0);
			hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			printf(" %8.2f", ms * 1e-3 * 2.4e9 / ((double)iters * en.per_iter * wps));
		}
		printf("\n");
		fflush(stdout);
	}
	return 0;
}
