// Stable LSD radix sort of (key, value) pairs on the device: 4-bit digits, histogram / scan / scatter per pass.
// Equal keys keep their input order, so whatever is built from the sorted order is the same on every run.
// Used by the prepass of the default scan (sp_cylm_scan.h: triangles of a class ordered by cylinder radius) and by the
// opt-in BVH build (sp_bvh_build.h: Morton keys).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sp {

// ---- LSD radix sort of (key, value) pairs, 4 bits per pass, stable.  A block owns kRsPerBlock consecutive elements.
constexpr uint32_t kRsPerBlock = 2048;

__global__ void __launch_bounds__(256) k_rs_hist(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t nblocks, uint32_t* __restrict__ hist) {
	__shared__ uint32_t h[16];
	if (threadIdx.x < 16) h[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t base = blockIdx.x * kRsPerBlock;
	for (uint32_t r = 0; r < kRsPerBlock / 256u; ++r) {
		const uint32_t e = base + r * 256u + threadIdx.x;
		if (e < n) atomicAdd(&h[(keys[e] >> shift) & 15u], 1u);
	}
	__syncthreads();
	if (threadIdx.x < 16) hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];      // digit-major: one running sum over the whole array
}

__global__ void __launch_bounds__(256) k_rs_scan(uint32_t* __restrict__ hist, uint32_t total) {   // exclusive prefix over 16 * nblocks counts
	__shared__ uint32_t part[256];
	const uint32_t tid = threadIdx.x, per = (total + 255u) / 256u;
	const uint32_t lo = tid * per < total ? tid * per : total, hi = lo + per < total ? lo + per : total;
	uint32_t sum = 0;
	for (uint32_t b = lo; b < hi; ++b) sum += hist[b];
	part[tid] = sum;
	__syncthreads();
	if (tid == 0) { uint32_t run = 0; for (int j = 0; j < 256; ++j) { const uint32_t v = part[j]; part[j] = run; run += v; } }
	__syncthreads();
	uint32_t run = part[tid];
	for (uint32_t b = lo; b < hi; ++b) { const uint32_t v = hist[b]; hist[b] = run; run += v; }
}

__global__ void __launch_bounds__(256) k_rs_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, uint32_t shift,
                                                   uint32_t nblocks, const uint32_t* __restrict__ hist, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
	__shared__ uint32_t run[16];          // elements of each digit this block has placed so far
	__shared__ uint32_t wcnt[4][16];
	const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63u;
	if (tid < 16) run[tid] = hist[(size_t)tid * nblocks + blockIdx.x];
	const uint32_t base = blockIdx.x * kRsPerBlock;
	for (uint32_t r = 0; r < kRsPerBlock / 256u; ++r) {
		__syncthreads();
		const uint32_t e = base + r * 256u + tid;
		const bool ok = e < n;
		const uint32_t key = ok ? keys[e] : 0u, val = ok ? vals[e] : 0u;
		const int dig = ok ? (int)((key >> shift) & 15u) : -1;
		uint32_t rank = 0;
#pragma unroll
		for (int d = 0; d < 16; ++d) {
			const unsigned long long m = __ballot(dig == d);
			if (dig == d) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
			if (lane == 0) wcnt[wv][d] = (uint32_t)__popcll(m);
		}
		__syncthreads();
		if (ok) {
			uint32_t pos = run[dig] + rank;
			for (uint32_t v = 0; v < wv; ++v) pos += wcnt[v][dig];
			keys_out[pos] = key; vals_out[pos] = val;
		}
		__syncthreads();
		if (tid < 16) run[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
	}
}

} // namespace sp
