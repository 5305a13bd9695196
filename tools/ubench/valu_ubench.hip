// valu_ubench.hip -- developer micro-benchmark: VALU issue cost per wave64 instruction on
// gfx950, per opcode, at 1/2/4/8 waves per SIMD.  Calibrates the compute roofline used in
// DESIGN.md (the FP32 "157.3 TFLOP/s" peak needs packed FMA; plain ops have a lower ceiling).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, ASM)                                                                  \
	__global__ __launch_bounds__(256) void k_##NAME(float *out, int iters, float b, float c) \
	{                                                                                      \
		float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
		float2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6}; \
		float2 pb = {b, b}, pc = {c, c};                                                    \
		for (int i = 0; i < iters; ++i)                                                     \
		{                                                                                  \
			ASM                                                                            \
		}                                                                                  \
		out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y; \
	}

#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define A_MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define A_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a##i));
#define A_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a##i));
#define A_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##i));
#define A_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
#define A_RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a##i));
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a##i) : "v"(b));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : );
#define A_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a##i), "v"(b) : "vcc");
#define A_CMPS(i) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a##i), "v"(b) : "s20", "s21");
#define A_DIVSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a##i) : "v"(b) : "vcc");
#define A_DIVFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c) : "vcc");
#define A_DIVFIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##i) : "v"(pb), "v"(pc));
#define A_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(pb));
#define A_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##i) : "v"(pc));
#define A_FMAK(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_CVT(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##i));
#define A_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a##i));
#define A_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_SUBREV_S(i) asm volatile("v_subrev_f32 %0, s4, %0" : "+v"(a##i));
#define A_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
#define A_FRACT(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a##i));
#define A_CNDMASK_S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a##i) : "v"(b));
#define A_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define A_SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define A_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define A_FMAMK(i) asm volatile("v_fmamk_f32 %0, %0, 0x3f8ccccd, %1" : "+v"(a##i) : "v"(c));
#define A_ADD_LIT(i) asm volatile("v_add_f32 %0, 0x3f8ccccd, %0" : "+v"(a##i));
#define A_MUL_SGPR(i) asm volatile("v_mul_f32 %0, s4, %0" : "+v"(a##i));
#define A_ADD_ABS(i) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(a##i) : "v"(c));
#define A_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_TRUNC(i) asm volatile("v_trunc_f32 %0, %0" : "+v"(a##i));
#define A_CVT_I(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a##i));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_FMA_MIN(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_min_f32 %3, %3, %2" : "+v"(a##i), "+v"(p##i.x) : "v"(b), "v"(c));
#define A_FMA_SQRT(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_sqrt_f32 %3, %3" : "+v"(a##i), "+v"(p##i.x) : "v"(b), "v"(c));
#define A_DEP(i)  /* dependent chain on a0 only */

KERNEL(fma, REP8(A_FMA))
KERNEL(fmac, REP8(A_FMAK))
KERNEL(mul, REP8(A_MUL))
KERNEL(add, REP8(A_ADD))
KERNEL(min, REP8(A_MIN))
KERNEL(max3, REP8(A_MAX3))
KERNEL(floor, REP8(A_FLOOR))
KERNEL(rndne, REP8(A_RNDNE))
KERNEL(fract, REP8(A_FRACT))
KERNEL(sqrt, REP8(A_SQRT))
KERNEL(rcp, REP8(A_RCP))
KERNEL(rsq, REP8(A_RSQ))
KERNEL(exp, REP8(A_EXP))
KERNEL(mov, REP8(A_MOV))
KERNEL(cndmask, REP8(A_CNDMASK))
KERNEL(cmp_vcc, REP8(A_CMP))
KERNEL(cmp_sgpr, REP8(A_CMPS))
KERNEL(div_scale, REP8(A_DIVSCALE))
KERNEL(div_fmas, REP8(A_DIVFMAS))
KERNEL(div_fixup, REP8(A_DIVFIXUP))
KERNEL(pk_fma, REP8(A_PKFMA))
KERNEL(pk_mul, REP8(A_PKMUL))
KERNEL(pk_add, REP8(A_PKADD))
KERNEL(cvt_f32_u32, REP8(A_CVT))
KERNEL(lshl, REP8(A_LSHL))
KERNEL(add_u32, REP8(A_ADDU))
KERNEL(subrev_sgpr, REP8(A_SUBREV_S))
KERNEL(cndmask_sgpr, REP8(A_CNDMASK_S))
KERNEL(max, REP8(A_MAX))
KERNEL(sub, REP8(A_SUB))
KERNEL(and, REP8(A_AND))
KERNEL(fmamk, REP8(A_FMAMK))
KERNEL(add_lit, REP8(A_ADD_LIT))
KERNEL(mul_sgpr, REP8(A_MUL_SGPR))
KERNEL(add_abs, REP8(A_ADD_ABS))
KERNEL(min3, REP8(A_MIN3))
KERNEL(med3, REP8(A_MED3))
KERNEL(trunc, REP8(A_TRUNC))
KERNEL(cvt_i32, REP8(A_CVT_I))
KERNEL(mul_lo_u32, REP8(A_MULLO))
KERNEL(fma_min_pair, REP8(A_FMA_MIN))
KERNEL(fma3_sqrt, REP8(A_FMA_SQRT))

// dependent chain: every instruction depends on the previous one
__global__ __launch_bounds__(256) void k_fma_dep(float *out, int iters, float b, float c)
{
	float a = threadIdx.x;
	for (int i = 0; i < iters; ++i)
	{
		asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
					 "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
					 : "+v"(a) : "v"(b), "v"(c));
	}
	out[blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_mul_dep(float *out, int iters, float b, float c)
{
	float a = threadIdx.x;
	for (int i = 0; i < iters; ++i)
	{
		asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n"
					 "v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n"
					 : "+v"(a) : "v"(b), "v"(c));
	}
	out[blockIdx.x * 256 + threadIdx.x] = a;
}

typedef void (*kfn)(float *, int, float, float);
struct Entry { const char *name; kfn fn; };

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const double clock_hz = prop.clockRate * 1e3;
	printf("device %s, %d CUs, clockRate %.0f MHz\n", prop.name, cus, clock_hz / 1e6);
	float *out;
	hipMalloc(&out, sizeof(float) * 256 * cus * 8);
	std::vector<Entry> entries = {
#define E(n) {#n, k_##n},
		E(fma) E(fmac) E(mul) E(add) E(min) E(max3) E(floor) E(rndne) E(fract) E(mov) E(cndmask) E(cmp_vcc) E(cmp_sgpr) E(cvt_f32_u32) E(lshl) E(add_u32)
		E(subrev_sgpr) E(cndmask_sgpr) E(max) E(sub) E(and) E(fmamk) E(add_lit) E(mul_sgpr) E(add_abs) E(min3) E(med3) E(trunc) E(cvt_i32) E(mul_lo_u32) E(pk_fma) E(pk_mul) E(pk_add) E(div_scale) E(div_fmas) E(div_fixup) E(sqrt) E(rcp) E(rsq) E(exp) E(fma_dep) E(mul_dep) E(fma_min_pair) E(fma3_sqrt)
	};
	const int iters = 40000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	printf("%-14s %10s %10s %10s %10s   (cycles per wave64 instruction per SIMD at 2.4 GHz; waves/SIMD = 1,2,4,8)\n", "op", "w1", "w2", "w4", "w8");
	for (auto &en : entries)
	{
		printf("%-14s", en.name);
		for (int wps : {1, 2, 4, 8})
		{
			const int blocks = cus * wps; // one 256-thread block = 4 waves = 1 wave per SIMD
			hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
			hipDeviceSynchronize();
			hipEventRecord(e0);
			hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			const int per_iter = (std::string(en.name) == "fma_min_pair") ? 16 : (std::string(en.name) == "fma3_sqrt") ? 32 : 8;
			const double insts_per_simd = (double)iters * per_iter * wps; // wave-instructions issued on one SIMD
			const double cyc = ms * 1e-3 * 2.4e9 / insts_per_simd;
			printf(" %10.2f", cyc);
		}
		printf("\n");
		fflush(stdout);
	}
	return 0;
}
