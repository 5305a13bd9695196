// bank_ubench.hip -- developer micro-benchmark: does the VGPR operand placement of a VALU instruction change its issue cost on gfx950?
// Fixed register numbers; sources in one bank (register number mod 4 equal) against sources spread over banks; destination in or out of
// a source's bank.  Prints cycles per wave64 instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD (as tools/ubench/valu_ubench.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define INIT "v_mov_b32 v4, %1\n v_mov_b32 v5, %1\n v_mov_b32 v6, %1\n v_mov_b32 v7, %1\n v_mov_b32 v8, %2\n v_mov_b32 v9, %2\n v_mov_b32 v10, %2\n v_mov_b32 v11, %2\n" \
	"v_mov_b32 v12, %2\n v_mov_b32 v13, %2\n v_mov_b32 v14, %2\n v_mov_b32 v15, %2\n v_mov_b32 v16, %1\n v_mov_b32 v17, %1\n v_mov_b32 v18, %1\n v_mov_b32 v19, %1\n" \
	"v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
#define CLOBBER "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"

#define KERNEL(NAME, BODY)                                                                       \
	__global__ __launch_bounds__(256) void k_##NAME(float *out, int iters, float b, float c)    \
	{                                                                                            \
		float r;                                                                                 \
		asm volatile(INIT : "=v"(r) : "v"(b), "v"(c) : CLOBBER);                                   \
		for (int i = 0; i < iters; ++i) asm volatile(BODY : : : CLOBBER);                          \
		asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v25\n v_add_f32 %0, %0, v26\n v_add_f32 %0, %0, v27" : "=v"(r) : : CLOBBER); \
		out[blockIdx.x * 256 + threadIdx.x] = r;                                                 \
	}

// eight independent instructions per iteration, destinations v20..v27
#define R8(OP, A, B, C) OP " v20, " A ", " B C "\n" OP " v21, " A ", " B C "\n" OP " v22, " A ", " B C "\n" OP " v23, " A ", " B C "\n" OP " v24, " A ", " B C "\n" OP " v25, " A ", " B C "\n" OP " v26, " A ", " B C "\n" OP " v27, " A ", " B C "\n"
KERNEL(fma_3same, R8("v_fma_f32", "v4", "v8", ", v12"))
KERNEL(fma_2same, R8("v_fma_f32", "v4", "v8", ", v13"))
KERNEL(fma_spread, R8("v_fma_f32", "v4", "v9", ", v14"))
KERNEL(fma_one_reg, R8("v_fma_f32", "v4", "v4", ", v4"))
KERNEL(mul_same, R8("v_mul_f32", "v4", "v8", ""))
KERNEL(mul_spread, R8("v_mul_f32", "v4", "v9", ""))
KERNEL(add_same, R8("v_add_f32", "v4", "v8", ""))
KERNEL(add_spread, R8("v_add_f32", "v4", "v9", ""))
KERNEL(max_same, R8("v_max_f32", "v4", "v8", ""))
KERNEL(max_spread, R8("v_max_f32", "v4", "v9", ""))
// accumulating forms: the destination is also a source
#define ACC8(OP, B, C) OP " v20, v20, " B C "\n" OP " v21, v21, " B C "\n" OP " v22, v22, " B C "\n" OP " v23, v23, " B C "\n" OP " v24, v24, " B C "\n" OP " v25, v25, " B C "\n" OP " v26, v26, " B C "\n" OP " v27, v27, " B C "\n"
KERNEL(fmaacc_same, ACC8("v_fma_f32", "v4", ", v8"))     // v20 (bank 0), v4 (0), v8 (0) for dst 20, 24; mixed for the others
KERNEL(fmaacc_spread, ACC8("v_fma_f32", "v5", ", v10"))

// which pair matters: banks of (src0, src1, src2)
KERNEL(fma_s0s1, R8("v_fma_f32", "v4", "v8", ", v13"))   // (0, 0, 1)
KERNEL(fma_s1s2, R8("v_fma_f32", "v4", "v9", ", v13"))   // (0, 1, 1)
KERNEL(fma_s0s2, R8("v_fma_f32", "v4", "v9", ", v12"))   // (0, 1, 0)
KERNEL(max3_3same, R8("v_max3_f32", "v4", "v8", ", v12"))
KERNEL(max3_spread, R8("v_max3_f32", "v4", "v9", ", v14"))
// v_fmac_f32 vD, vA, vB: D = A * B + D -- destinations all in bank 0 (v20, v24, ... would need more registers: use 20, 24 only, four times)
#define FMAC8(A, B) "v_fmac_f32 v20, " A ", " B "\n v_fmac_f32 v24, " A ", " B "\n v_fmac_f32 v20, " A ", " B "\n v_fmac_f32 v24, " A ", " B "\n v_fmac_f32 v20, " A ", " B "\n v_fmac_f32 v24, " A ", " B "\n v_fmac_f32 v20, " A ", " B "\n v_fmac_f32 v24, " A ", " B "\n"
KERNEL(fmac_dst_a_b, FMAC8("v4", "v8"))     // (0, 0) + dst 0: all three in one bank
KERNEL(fmac_dst_b, FMAC8("v5", "v8"))       // src1 and dst in one bank
KERNEL(fmac_dst_a, FMAC8("v4", "v9"))       // src0 and dst in one bank
KERNEL(fmac_spread, FMAC8("v5", "v10"))
#define FMA2DST(A, B, C) "v_fma_f32 v20, " A ", " B ", " C "\n v_fma_f32 v24, " A ", " B ", " C "\n v_fma_f32 v20, " A ", " B ", " C "\n v_fma_f32 v24, " A ", " B ", " C "\n v_fma_f32 v20, " A ", " B ", " C "\n v_fma_f32 v24, " A ", " B ", " C "\n v_fma_f32 v20, " A ", " B ", " C "\n v_fma_f32 v24, " A ", " B ", " C "\n"
KERNEL(fma_2dst_spread, FMA2DST("v5", "v10", "v15"))  // the same two destinations as the fmac rows, sources spread

typedef void (*kfn)(float *, int, float, float);
struct Entry { const char *name; kfn fn; };

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	printf("device %s, %d CUs\n", prop.name, cus);
	float *out;
	hipMalloc(&out, sizeof(float) * 256 * cus * 8);
	std::vector<Entry> entries = {
#define E(n) {#n, k_##n},
		E(fma_3same) E(fma_2same) E(fma_spread) E(fma_one_reg) E(mul_same) E(mul_spread) E(add_same) E(add_spread) E(max_same) E(max_spread) E(fmaacc_same) E(fmaacc_spread) E(fma_s0s1) E(fma_s1s2) E(fma_s0s2) E(max3_3same) E(max3_spread) E(fmac_dst_a_b) E(fmac_dst_b) E(fmac_dst_a) E(fmac_spread) E(fma_2dst_spread)
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
			const int blocks = cus * wps;
			hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
			hipDeviceSynchronize();
			hipEventRecord(e0);
			hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			printf(" %10.2f", ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wps));
		}
		printf("\n");
		fflush(stdout);
	}
	return 0;
}
