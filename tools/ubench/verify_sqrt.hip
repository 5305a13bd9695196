// verify_sqrt.hip -- exhaustive check (all 2^32 bit patterns) of candidate fast
// correctly-rounded sqrt / constant-division sequences against hipcc's IEEE lowering.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float core_sqrt(float x, float clampv)
{
	float t = __builtin_fmaxf(x, clampv);
	float y = __builtin_amdgcn_rsqf(t);
	float g = x * y;
	float h = 0.5f * y;
	float r = __builtin_fmaf(-h, g, 0.5f);
	g = __builtin_fmaf(g, r, g);
	h = __builtin_fmaf(h, r, h);
	float d = __builtin_fmaf(-g, g, x);
	return __builtin_fmaf(d, h, g);
}
// variant without the Goldschmidt step
__device__ __forceinline__ float core_sqrt_short(float x, float clampv)
{
	float t = __builtin_fmaxf(x, clampv);
	float y = __builtin_amdgcn_rsqf(t);
	float g = x * y;
	float h = 0.5f * y;
	float d = __builtin_fmaf(-g, g, x);
	return __builtin_fmaf(d, h, g);
}

// mismatch histogram by biased exponent of x (sign folded: +256 for negative)
__global__ void k_sqrt(unsigned long long *hist, int variant, float clampv)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride)
	{
		float x = __uint_as_float((uint32_t)u);
		float ref = __builtin_sqrtf(x);
		float got = variant == 0 ? core_sqrt(x, clampv) : core_sqrt_short(x, clampv);
		bool same = (__float_as_uint(ref) == __float_as_uint(got)) || (ref != ref && got != got);
		if (!same) atomicAdd(&hist[((uint32_t)u >> 23) & 0x1ff], 1ull);
	}
}

// a / C for a compile-time constant C:  q = a*rc; r = fma(-C, q, a); q' = fma(r, rc, q)
template <int WHICH>
__device__ __forceinline__ float div_c(float a, float C)
{
	const float rc = 1.0f / C;
	float q = a * rc;
	float r = __builtin_fmaf(-C, q, a);
	return __builtin_fmaf(r, rc, q);
}
__global__ void k_div(unsigned long long *hist, float C)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride)
	{
		float a = __uint_as_float((uint32_t)u);
		float ref = a / C;
		float got = div_c<0>(a, C);
		bool same = (__float_as_uint(ref) == __float_as_uint(got)) || (ref != ref && got != got);
		if (!same) atomicAdd(&hist[((uint32_t)u >> 23) & 0x1ff], 1ull);
	}
}

static void report(const char *name, unsigned long long *d_hist)
{
	unsigned long long h[512];
	hipMemcpy(h, d_hist, sizeof h, hipMemcpyDeviceToHost);
	unsigned long long total = 0;
	for (int i = 0; i < 512; ++i) total += h[i];
	printf("%-40s mismatches %llu", name, total);
	if (total)
	{
		printf("  [sign|exp: count]");
		int shown = 0;
		for (int i = 0; i < 512 && shown < 24; ++i)
			if (h[i]) { printf(" %s%d:%llu", i >= 256 ? "-" : "+", i & 255, h[i]); ++shown; }
	}
	printf("\n");
	hipMemset(d_hist, 0, sizeof h);
}

int main()
{
	unsigned long long *d_hist;
	hipMalloc(&d_hist, 512 * sizeof(unsigned long long));
	hipMemset(d_hist, 0, 512 * sizeof(unsigned long long));
	const float clamps[] = {0x1p-126f, 0x1p-96f};
	for (float c : clamps)
		for (int v = 0; v < 2; ++v)
		{
			hipLaunchKernelGGL(k_sqrt, dim3(4096), dim3(256), 0, 0, d_hist, v, c);
			hipDeviceSynchronize();
			char name[96];
			snprintf(name, sizeof name, "sqrt variant %d clamp %g", v, c);
			report(name, d_hist);
		}
	const float consts[] = {20.f, 3.f, 10.f, 15.f, 2.01f, 0.035355339059327376f /* sqrt2*0.1/4 */, 255.f, 7.f, 289.f, 0.1f, 0.4f, 5.f, 6.28318530717958647f, 8.f};
	for (float c : consts)
	{
		hipLaunchKernelGGL(k_div, dim3(4096), dim3(256), 0, 0, d_hist, c);
		hipDeviceSynchronize();
		char name[96];
		snprintf(name, sizeof name, "a / %.9g via rc + 2 fma", c);
		report(name, d_hist);
	}
	return 0;
}
