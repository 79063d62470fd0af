// Device-side build of the opt-in linear BVH (sp_bvh.h, SPHIP_FLAG_ACCEL).
// Nothing leaves the GPU: scene box -> "big" classification -> Morton keys -> LSD radix sort (own kernels, 4-bit digits,
// stable, so equal keys stay in index order and the structure is the same on every run) -> leaf records and boxes ->
// bottom-up refit, one launch per level.  Replaces round 1's D2H + std::sort on the host.
#pragma once

#include "sp_kernels.h"
#include "sp_radix_sort.h"

namespace sp {

// meta words (device): [0..2] scene lo.xyz, [3..5] hi.xyz as ordered uints; [6] triangles wider than 1/4 of the scene,
// [7] wider than 1/2; [8] n_big actually used; [9] n_tree; [10] threshold bits (float)
constexpr uint32_t kBvhMaxBig = 256;

SP_DEV uint32_t f2ord(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
SP_DEV float ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__global__ void __launch_bounds__(256) k_bvh_meta_init(uint32_t* __restrict__ meta) {
	const uint32_t t = threadIdx.x;
	if (t < 3) meta[t] = 0xffffffffu;             // min over ordered uints
	else if (t < 6) meta[t] = 0u;                 // max
	else if (t < 16) meta[t] = 0u;
}

// ---- pass 1: scene box over the finite vertex coordinates
__global__ void __launch_bounds__(256) k_bvh_box(const float* __restrict__ tris, uint32_t n, uint32_t* __restrict__ meta) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	uint32_t lo[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, hi[3] = { 0u, 0u, 0u };
	if (i < n) {
		const float* t = tris + (size_t)i * 12;
#pragma unroll
		for (int k = 0; k < 9; ++k) {
			const float v = t[k];
			if ((v - v) == 0.0f) { const uint32_t o = f2ord(v); lo[k % 3] = o < lo[k % 3] ? o : lo[k % 3]; hi[k % 3] = o > hi[k % 3] ? o : hi[k % 3]; }
		}
	}
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		uint32_t l = lo[a], h = hi[a];
		for (int off = 32; off > 0; off >>= 1) { const uint32_t l2 = __shfl_xor(l, off, 64), h2 = __shfl_xor(h, off, 64); l = l2 < l ? l2 : l; h = h2 > h ? h2 : h; }
		if ((threadIdx.x & 63u) == 0) { if (l != 0xffffffffu) atomicMin(meta + a, l); if (h != 0u) atomicMax(meta + 3 + a, h); }
	}
}

SP_DEV float tri_extent(const float* __restrict__ t) {         // largest side of the triangle's box; NaN if a coordinate is
	float e = 0.0f;
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		const float v0 = t[a], v1 = t[3 + a], v2 = t[6 + a];
		const float d = fmaxf(v0, fmaxf(v1, v2)) - fminf(v0, fminf(v1, v2));
		e = (d > e || d != d) ? d : e;
	}
	return e;
}

SP_DEV void scene_box(const uint32_t* __restrict__ meta, float (&lo)[3], float (&ext)[3], float& max_ext, float& scale) {
	max_ext = 0.0f; scale = 0.0f;
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		float l = ord2f(meta[a]), h = ord2f(meta[3 + a]);
		if (meta[a] == 0xffffffffu || !(h >= l)) { l = 0.0f; h = 0.0f; }           // no finite coordinate at all
		lo[a] = l; ext[a] = h - l;
		max_ext = fmaxf(max_ext, ext[a]);
		scale = fmaxf(scale, fmaxf(fabsf(l), fabsf(h)));
	}
}

// ---- pass 2: how many triangles span more than 1/4 (1/2) of the scene
__global__ void __launch_bounds__(256) k_bvh_count_big(const float* __restrict__ tris, uint32_t n, uint32_t* __restrict__ meta) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	float lo[3], ext[3], max_ext, scale;
	scene_box(meta, lo, ext, max_ext, scale);
	const float e = i < n ? tri_extent(tris + (size_t)i * 12) : 0.0f;
	const int c4 = __syncthreads_count(i < n && !(e <= 0.25f * max_ext));      // also counts NaN extents
	const int c2 = __syncthreads_count(i < n && !(e <= 0.5f * max_ext));
	if (threadIdx.x == 0) { if (c4) atomicAdd(meta + 6, (uint32_t)c4); if (c2) atomicAdd(meta + 7, (uint32_t)c2); }
}

// The big triangles (room walls, ground planes) would put scene-sized boxes on whole root-to-leaf paths: they stay out of
// the tree and are tested for every ray.  At most kBvhMaxBig of them: the threshold is raised until that holds.
SP_DEV float big_threshold(const uint32_t* __restrict__ meta, float max_ext) {
	return meta[6] <= kBvhMaxBig ? 0.25f * max_ext : (meta[7] <= kBvhMaxBig ? 0.5f * max_ext : __builtin_inff());
}

SP_DEV uint32_t morton10(float x) {   // x in [0,1): spread 10 bits to every third position
	uint32_t v = (uint32_t)fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
	v = (v | (v << 16)) & 0x030000FFu;
	v = (v | (v << 8)) & 0x0300F00Fu;
	v = (v | (v << 4)) & 0x030C30C3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}

// ---- pass 3: sort key per triangle: the Morton code of its centroid; big triangles get 0xffffffff (sorted behind the tree's)
__global__ void __launch_bounds__(256) k_bvh_keys(const float* __restrict__ tris, uint32_t n, uint32_t* __restrict__ meta,
                                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	float lo[3], ext[3], max_ext, scale;
	scene_box(meta, lo, ext, max_ext, scale);
	const float thr = big_threshold(meta, max_ext);
	bool big = false;
	if (i < n) {
		const float* t = tris + (size_t)i * 12;
		big = !(tri_extent(t) <= thr) && thr != __builtin_inff();
		uint32_t code = 0;
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			const float cen = (t[a] + t[3 + a] + t[6 + a]) * (1.0f / 3.0f);
			const float u = ext[a] > 0.0f ? (cen - lo[a]) / ext[a] : 0.0f;
			code |= morton10((u - u) == 0.0f ? u : 0.0f) << a;
		}
		keys[i] = big ? 0xffffffffu : code;
		vals[i] = i;
	}
	const int nb = __syncthreads_count(big);
	if (threadIdx.x == 0 && nb) atomicAdd(meta + 8, (uint32_t)nb);
}

// ---- leaves: sorted position j < n_tree -> exact record j, leaf j/4; big triangles -> records 4*n_leaves + k
__global__ void __launch_bounds__(256) k_bvh_leaves(const float* __restrict__ tris, uint32_t n, const uint32_t* __restrict__ meta, const uint32_t* __restrict__ sorted,
                                                   uint32_t n_leaves, float4* __restrict__ nodes, float4* __restrict__ rec, int* __restrict__ idx) {
	const uint32_t leaf = blockIdx.x * 256u + threadIdx.x;
	if (leaf >= n_leaves) return;
	float slo[3], sext[3], max_ext, scale;
	scene_box(meta, slo, sext, max_ext, scale);
	const uint32_t n_tree = n - meta[8];
	float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, hi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
	for (uint32_t k = 0; k < 4; ++k) {
		const uint32_t j = leaf * 4u + k;
		float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0;
		int orig = -1;
		if (j < n_tree) {
			orig = (int)sorted[j];
			const float* s = tris + (size_t)orig * 12;
			r0 = make_float4(s[0], s[1], s[2], s[3] - s[0]);                    // e1, e2: the reference's float subtractions (geom.h:200-201)
			r1 = make_float4(s[4] - s[1], s[5] - s[2], s[6] - s[0], s[7] - s[1]);
			r2 = make_float4(s[8] - s[2], 0.0f, 0.0f, 0.0f);
#pragma unroll
			for (int c = 0; c < 9; ++c) { const float v = s[c]; if ((v - v) == 0.0f) { lo[c % 3] = fminf(lo[c % 3], v); hi[c % 3] = fmaxf(hi[c % 3], v); } }
		}
		rec[(size_t)j * 3] = r0; rec[(size_t)j * 3 + 1] = r1; rec[(size_t)j * 3 + 2] = r2;
		idx[j] = orig;
	}
	if (hi[0] >= lo[0]) {        // inflate (the slab test runs in float: keep every geometric hit inside); an empty leaf keeps its inverted box
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			const float pad = 1e-5f * fmaxf(fmaxf(fabsf(lo[a]), fabsf(hi[a])), hi[a] - lo[a]) + 1e-6f * scale + 1e-30f;
			lo[a] -= pad; hi[a] += pad;
		}
	}
	nodes[(size_t)(n_leaves + leaf) * 2] = make_float4(lo[0], lo[1], lo[2], hi[0]);
	nodes[(size_t)(n_leaves + leaf) * 2 + 1] = make_float4(hi[1], hi[2], 0.0f, 0.0f);
}

__global__ void __launch_bounds__(256) k_bvh_bigs(const float* __restrict__ tris, uint32_t n, const uint32_t* __restrict__ meta, const uint32_t* __restrict__ sorted,
                                                 uint32_t n_leaves, float4* __restrict__ rec, int* __restrict__ idx) {
	const uint32_t k = threadIdx.x;                      // kBvhMaxBig == 256 == one workgroup
	const uint32_t n_big = meta[8], n_tree = n - n_big;
	const size_t j = (size_t)4 * n_leaves + k;
	float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0;
	int orig = -1;
	if (k < n_big) {
		orig = (int)sorted[n_tree + k];
		const float* s = tris + (size_t)orig * 12;
		r0 = make_float4(s[0], s[1], s[2], s[3] - s[0]);
		r1 = make_float4(s[4] - s[1], s[5] - s[2], s[6] - s[0], s[7] - s[1]);
		r2 = make_float4(s[8] - s[2], 0.0f, 0.0f, 0.0f);
	}
	rec[j * 3] = r0; rec[j * 3 + 1] = r1; rec[j * 3 + 2] = r2;
	idx[j] = orig;
}

// ---- refit one level of the complete tree (heap order): nodes [first, 2*first)
__global__ void __launch_bounds__(256) k_bvh_refit(float4* __restrict__ nodes, uint32_t first) {
	const uint32_t node = first + blockIdx.x * 256u + threadIdx.x;
	if (node >= 2u * first) return;
	const float4 l0 = nodes[(size_t)4 * node], l1 = nodes[(size_t)4 * node + 1], r0 = nodes[(size_t)4 * node + 2], r1 = nodes[(size_t)4 * node + 3];
	nodes[(size_t)2 * node] = make_float4(fminf(l0.x, r0.x), fminf(l0.y, r0.y), fminf(l0.z, r0.z), fmaxf(l0.w, r0.w));
	nodes[(size_t)2 * node + 1] = make_float4(fmaxf(l1.x, r1.x), fmaxf(l1.y, r1.y), 0.0f, 0.0f);
}

} // namespace sp
