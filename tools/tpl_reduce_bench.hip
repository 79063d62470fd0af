// Experiment (not part of the library): the closest-hit scan in the layout BASELINE.json's north_star sketches --
// TRIANGLE PER LANE, ray wave-uniform, nearest hit picked by a WAVEFRONT-WIDE MIN-REDUCE -- measured against the
// ray-per-lane exact scan the library uses (same strict arithmetic, sp_device_math.h), on the same inputs.
//
// Each wave owns RB rays (kept in SGPRs) and sweeps the whole triangle array 64 triangles at a time (one per lane,
// coalesced 48-B records).  Per lane and ray a packed key (float_bits(d) << 32 | index) keeps the running minimum
// over the tiles (d > 0, so float bits order like unsigned ints; the index in the low half gives the reference's
// "lowest index wins ties" rule, cpu_renderer.cpp:44).  After the sweep one 6-step wave min-reduce per ray picks the
// nearest hit.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -Ispath_amd/csrc -o build/tpl_reduce_bench tools/tpl_reduce_bench.hip
#include "sp_device_math.h"

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using namespace sp;

constexpr int RB = 8;   // rays per wave (6 SGPRs each)

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const unsigned long long o = __shfl_xor(v, off, 64);
		v = o < v ? o : v;
	}
	return v;
}

// triangle per lane + wave min-reduce
__global__ void __launch_bounds__(256) k_tpl(const float* __restrict__ rays, const float4* __restrict__ scan, uint32_t n_tris_padded,
                                              uint32_t n_rays, int* __restrict__ out_idx, float* __restrict__ out_d) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
	const uint32_t r0 = __builtin_amdgcn_readfirstlane(wave * RB);          // wave-uniform: ray data arrives through the scalar path
	if (r0 >= n_rays) return;
	f3 o[RB], d[RB];
#pragma unroll
	for (int r = 0; r < RB; ++r) {
		const uint32_t k = (r0 + r < n_rays) ? r0 + r : n_rays - 1;
		const float* p = rays + (size_t)k * 6;
		o[r] = mk3(p[0], p[1], p[2]); d[r] = mk3(p[3], p[4], p[5]);
	}
	unsigned long long key[RB];
#pragma unroll
	for (int r = 0; r < RB; ++r) key[r] = ~0ull;
	for (uint32_t base = 0; base < n_tris_padded; base += 64) {
		const uint32_t j = base + lane;
		const float4 q0 = scan[3 * j + 0], q1 = scan[3 * j + 1], q2 = scan[3 * j + 2];   // 48 B per lane, consecutive lanes consecutive records
		const f3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
#pragma unroll
		for (int r = 0; r < RB; ++r) {
			const float dist = ray_tri_strict(o[r], d[r], v0, e1, e2);
			const unsigned long long k = dist > 0.0f ? (((unsigned long long)__float_as_uint(dist) << 32) | j) : ~0ull;
			key[r] = k < key[r] ? k : key[r];
		}
	}
#pragma unroll
	for (int r = 0; r < RB; ++r) {
		const unsigned long long m = wave_min_u64(key[r]);
		if (lane == 0 && r0 + r < n_rays) {
			const float dist = __uint_as_float((uint32_t)(m >> 32));
			const bool hit = (m != ~0ull) && (dist < kMaxDist);
			out_idx[r0 + r] = hit ? (int)(uint32_t)m : -1;
			out_d[r0 + r] = hit ? dist : kMaxDist;
		}
	}
}

// ray per lane, triangle broadcast from LDS tiles (what the library's rpl_lds does), for the same-binary comparison
__global__ void __launch_bounds__(256) k_rpl(const float* __restrict__ rays, const float4* __restrict__ scan, uint32_t n_tris_padded,
                                              uint32_t n_rays, int* __restrict__ out_idx, float* __restrict__ out_d) {
	__shared__ float4 sm[768];
	const uint32_t tid = threadIdx.x, k = blockIdx.x * 256u + tid;
	const uint32_t kk = k < n_rays ? k : n_rays - 1;
	const float* p = rays + (size_t)kk * 6;
	const f3 o = mk3(p[0], p[1], p[2]), d = mk3(p[3], p[4], p[5]);
	float bd = kMaxDist; int bi = -1;
	for (uint32_t base = 0; base < n_tris_padded; base += 256) {
		__syncthreads();
		sm[tid] = scan[3 * base + tid]; sm[256 + tid] = scan[3 * base + 256 + tid]; sm[512 + tid] = scan[3 * base + 512 + tid];
		__syncthreads();
#pragma unroll 4
		for (uint32_t j = 0; j < 256; ++j) {
			const float4 q0 = sm[3 * j], q1 = sm[3 * j + 1], q2 = sm[3 * j + 2];
			const float dist = ray_tri_strict(o, d, mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), mk3(q1.z, q1.w, q2.x));
			const bool take = dist > 0.0f && dist < bd;
			bd = take ? dist : bd; bi = take ? (int)(base + j) : bi;
		}
	}
	if (k < n_rays) { out_idx[k] = bi; out_d[k] = bd; }
}

int main(int argc, char** argv) {
	const uint32_t n_tris = argc > 1 ? (uint32_t)atoi(argv[1]) : 10000, n_rays = argc > 2 ? (uint32_t)atoi(argv[2]) : (1u << 20);
	const uint32_t n_pad = (n_tris + 255) / 256 * 256;
	std::vector<float> scan((size_t)n_pad * 12, 0.0f), rays((size_t)n_rays * 6);
	uint32_t s = 12345;
	auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
	for (uint32_t i = 0; i < n_tris; ++i) {            // small random triangles in a 3x2x3 box; records: v0, e1, e2
		float* t = &scan[(size_t)i * 12];
		const float c[3] = { rnd() * 3 - 1.5f, rnd() * 2 - 1.0f, rnd() * 3 - 1.5f };
		float v[9];
		for (int k = 0; k < 9; ++k) v[k] = c[k % 3] + (rnd() - 0.5f) * 0.1f;
		t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
		t[3] = v[3] - v[0]; t[4] = v[4] - v[1]; t[5] = v[5] - v[2];
		t[6] = v[6] - v[0]; t[7] = v[7] - v[1]; t[8] = v[8] - v[2];
	}
	for (uint32_t i = 0; i < n_rays; ++i) {
		float* r = &rays[(size_t)i * 6];
		r[0] = rnd() * 2 - 1; r[1] = rnd() - 0.5f; r[2] = rnd() * 2 - 1;
		float dx = rnd() - 0.5f, dy = rnd() - 0.5f, dz = rnd() - 0.5f, l = sqrtf(dx * dx + dy * dy + dz * dz) + 1e-9f;
		r[3] = dx / l; r[4] = dy / l; r[5] = dz / l;
	}
	float *d_scan, *d_rays, *d_d[2]; int* d_i[2];
	CHECK(hipMalloc(&d_scan, scan.size() * 4)); CHECK(hipMalloc(&d_rays, rays.size() * 4));
	for (int k = 0; k < 2; ++k) { CHECK(hipMalloc(&d_d[k], n_rays * 4)); CHECK(hipMalloc(&d_i[k], n_rays * 4)); }
	CHECK(hipMemcpy(d_scan, scan.data(), scan.size() * 4, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float ms[2] = { 0, 0 };
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k_tpl, dim3((n_rays / RB * 64 + 255) / 256), dim3(256), 0, 0, d_rays, (const float4*)d_scan, n_pad, n_rays, d_i[0], d_d[0]);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms[0], e0, e1));
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k_rpl, dim3((n_rays + 255) / 256), dim3(256), 0, 0, d_rays, (const float4*)d_scan, n_pad, n_rays, d_i[1], d_d[1]);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms[1], e0, e1));
	}
	std::vector<int> hi[2]; std::vector<float> hd[2];
	for (int k = 0; k < 2; ++k) {
		hi[k].resize(n_rays); hd[k].resize(n_rays);
		CHECK(hipMemcpy(hi[k].data(), d_i[k], n_rays * 4, hipMemcpyDeviceToHost));
		CHECK(hipMemcpy(hd[k].data(), d_d[k], n_rays * 4, hipMemcpyDeviceToHost));
	}
	size_t diff = 0, hits = 0;
	for (uint32_t i = 0; i < n_rays; ++i) { diff += (hi[0][i] != hi[1][i]) || memcmp(&hd[0][i], &hd[1][i], 4) != 0; hits += hi[1][i] >= 0; }
	const double tests = (double)n_rays * n_tris;
	printf("n_tris %u, n_rays %u, hit rate %.3f, results differ on %zu rays\n", n_tris, n_rays, (double)hits / n_rays, diff);
	printf("triangle-per-lane + wave u64 min-reduce (north_star layout): %8.2f ms  %.3f T tests/s\n", ms[0], tests / ms[0] / 1e9);
	printf("ray-per-lane, LDS-broadcast triangles (library's exact scan):  %8.2f ms  %.3f T tests/s\n", ms[1], tests / ms[1] / 1e9);
	return diff ? 2 : 0;
}
