// Exhaustive check of cheap correctly-rounded reciprocal candidates against IEEE division on gfx950.
// The reference computes f = 1.0/a (geom.h:206), an IEEE float divide; the compiler's expansion costs
// ~10 VALU instructions.  Candidates: v_rcp_f32 + Newton steps written with explicit FMAs.
// For every one of the 2^32 float bit patterns, compare each candidate with the device's IEEE divide;
// mismatches are histogrammed by biased exponent.  A strided sample of the device divide is also
// compared with the host's divide.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/rcp_check tools/rcp_check.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float cand1(float a) {   // rcp + 1 Newton step
	float r = __builtin_amdgcn_rcpf(a);
	float e = __builtin_fmaf(-a, r, 1.0f);
	return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float cand2(float a) {   // rcp + 2 Newton steps
	float r = __builtin_amdgcn_rcpf(a);
	float e = __builtin_fmaf(-a, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	e = __builtin_fmaf(-a, r, 1.0f);
	return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float cand3(float a) {   // the compiler's divide expansion without the scaling: 6 FMAs
	float r = __builtin_amdgcn_rcpf(a);
	float e = __builtin_fmaf(-a, r, 1.0f);
	const float r1 = __builtin_fmaf(e, r, r);
	const float r0 = __builtin_fmaf(-a, r1, 1.0f);
	const float q1 = __builtin_fmaf(r0, r1, r1);
	const float rr = __builtin_fmaf(-a, q1, 1.0f);
	return __builtin_fmaf(rr, r1, q1);
}

__device__ __forceinline__ bool same(float x, float y) {
	const uint32_t a = __float_as_uint(x), b = __float_as_uint(y);
	if (a == b) return true;
	return (x != x) && (y != y);
}

__global__ void k(unsigned long long* bad /* [3][512] */, float* sample /* 1<<20 */) {
	const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	for (uint64_t u = gid; u < (1ull << 32); u += stride) {
		const float a = __uint_as_float((uint32_t)u);
		const float ref = 1.0f / a;
		const uint32_t bucket = ((uint32_t)u >> 23);   // sign + exponent
		if (!same(cand1(a), ref)) atomicAdd(&bad[0 * 512 + bucket], 1ull);
		if (!same(cand2(a), ref)) atomicAdd(&bad[1 * 512 + bucket], 1ull);
		if (!same(cand3(a), ref)) atomicAdd(&bad[2 * 512 + bucket], 1ull);
		if ((u & 0xfffu) == 0x5a7u) sample[u >> 12] = ref;
	}
}

int main() {
	unsigned long long* d_bad; float* d_sample;
	CHECK(hipMalloc(&d_bad, 3 * 512 * 8)); CHECK(hipMemset(d_bad, 0, 3 * 512 * 8));
	CHECK(hipMalloc(&d_sample, (1u << 20) * 4));
	hipLaunchKernelGGL(k, dim3(256 * 16), dim3(256), 0, 0, d_bad, d_sample);
	CHECK(hipDeviceSynchronize());
	std::vector<unsigned long long> bad(3 * 512);
	std::vector<float> sample(1u << 20);
	CHECK(hipMemcpy(bad.data(), d_bad, 3 * 512 * 8, hipMemcpyDeviceToHost));
	CHECK(hipMemcpy(sample.data(), d_sample, (1u << 20) * 4, hipMemcpyDeviceToHost));
	long host_bad = 0;
	for (uint32_t i = 0; i < (1u << 20); ++i) {
		const uint32_t u = (i << 12) | 0x5a7u;
		float a; memcpy(&a, &u, 4);
		volatile float ref = 1.0f / a;
		float r = ref;
		if (memcmp(&r, &sample[i], 4) != 0 && !(r != r && sample[i] != sample[i])) {
			if (host_bad < 5) printf("device divide differs from host: a=%a dev=%a host=%a\n", a, sample[i], r);
			host_bad++;
		}
	}
	printf("device IEEE divide vs host divide on %u strided samples: %ld mismatches\n", 1u << 20, host_bad);
	const char* names[3] = { "rcp+1NR", "rcp+2NR", "rcp+6fma(unscaled div)" };
	for (int c = 0; c < 3; ++c) {
		unsigned long long total = 0;
		for (int b = 0; b < 512; ++b) total += bad[c * 512 + b];
		printf("%s: %llu mismatches of 2^32;", names[c], total);
		int lo = -1, hi = -1;
		// exponent range (positive sign) with zero mismatches
		for (int b = 1; b < 255; ++b) if (bad[c * 512 + b] == 0 && bad[c * 512 + 256 + b] == 0) { if (lo < 0) lo = b; hi = b; } else if (lo >= 0 && hi >= 0 && b > hi + 1) {}
		printf(" exact exponent buckets:");
		int run_lo = -1;
		for (int b = 0; b <= 256; ++b) {
			const bool ok = b < 256 && bad[c * 512 + b] == 0 && bad[c * 512 + 256 + b] == 0;
			if (ok && run_lo < 0) run_lo = b;
			if (!ok && run_lo >= 0) { printf(" [%d..%d]", run_lo, b - 1); run_lo = -1; }
		}
		printf("\n");
		for (int b = 0; b < 256; ++b) if (bad[c * 512 + b] || bad[c * 512 + 256 + b])
			if (b < 3 || b > 250 || (b % 16 == 0)) printf("    exp %3d: +%llu -%llu\n", b, bad[c * 512 + b], bad[c * 512 + 256 + b]);
	}
	return 0;
}
