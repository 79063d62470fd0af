// an acceleration structure behind the same C ABI -- OPT-IN (SPHIP_FLAG_ACCEL), never the
// default and never what bench.py's headline figure measures, because it changes the work definition: the reference
// has no acceleration structure (README.md:23) and tests every triangle.
//
// Structure: a linear BVH, built on the device (sp_bvh_build.h).  Triangles are sorted by the Morton code of their centroid, grouped four to a leaf, and
// the leaves (padded to a power of two) form a complete binary tree in heap order (root 1, children 2i and 2i+1), so
// parent / sibling / child links are index arithmetic and the tree is refitted bottom-up level by level.
// Traversal: one ray per lane, stackless with a bit trail (a set bit = "the far child of that level is still to do").
// Leaves run the SAME strict Moeller-Trumbore test as the brute-force scan (ray_tri_strict) on exact records
// that carry the triangle's ORIGINAL index, and hits are compared lexicographically by (d, original index), which
// is the reference's "first strictly smaller d wins, ascending index" rule (cpu_renderer.cpp:44) in any visiting order.
//
// Parity: the box test is conservative (boxes inflated, comparisons inclusive), so every GEOMETRIC hit the reference
// finds is found with the same index and the same distance bits.  What a bounding-volume cull cannot reproduce are
// the reference's noise accepts (a ray almost coplanar with a far-away triangle: a and s.h are both rounding noise and
// u, v land in [0,1] by chance -- DESIGN.md section 8).  tests/test_hip_accel.py measures how rare those are.
#pragma once

#include "sp_kernels.h"

namespace sp {

struct BvhArgs {
	const float4* nodes;     // 2 x float4 per node, heap order, index 0 unused: {lo.xyz, hi.x} {hi.yz, -, -}
	const float4* leaf_rec;  // 3 x float4 per sorted triangle: v0, e1, e2 (exact records), 4 triangles per leaf
	const int*    leaf_idx;  // original triangle index per sorted triangle, -1 for padding
	uint32_t n_leaves;       // power of two
	uint32_t first_leaf;     // == n_leaves (heap index of leaf 0)
	const uint32_t* meta;    // device words of the builder (sp_bvh_build.h); meta[8] = n_big: triangles too large for the tree (room walls,
	                         // ground planes): sorted records [4*n_leaves, 4*n_leaves + n_big) are tested for every ray before the walk,
	                         // which also gives the walk a tight cull distance from the start
};

// inclusive slab test; NaNs from 0 * inf drop out because v_min/v_max return the non-NaN operand
SP_DEV bool box_hit(const float4 a, const float4 b, f3 o, f3 inv, float tbest, float& tnear) {
	const float x0 = (a.x - o.x) * inv.x, x1 = (a.w - o.x) * inv.x;
	const float y0 = (a.y - o.y) * inv.y, y1 = (b.x - o.y) * inv.y;
	const float z0 = (a.z - o.z) * inv.z, z1 = (b.y - o.z) * inv.z;
	const float tmin = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
	const float tmax = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tbest));
	tnear = tmin;
	// (a.x <= a.w) is false for the inverted boxes of empty padding subtrees, which the slab arithmetic alone would
	// read as infinitely large
	return (a.x <= a.w) && (tmin <= tmax * 1.00001f + 1e-30f);
}

SP_DEV void scan_bvh(const BvhArgs& B, f3 o, f3 dir, int src, float& best_d, int& best_i, uint32_t* steps_out = nullptr, uint32_t* leaves_out = nullptr) {
	uint32_t n_steps = 0, n_leaves = 0;
	float bd = kMaxDist;
	int bi = -1;
	const f3 inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
	for (uint32_t j = 4u * B.n_leaves, e = 4u * B.n_leaves + B.meta[8]; j < e; ++j) {       // wave-uniform loop: scalar loads
		const int orig = B.leaf_idx[j];
		const float4 q0 = B.leaf_rec[3 * j], q1 = B.leaf_rec[3 * j + 1], q2 = B.leaf_rec[3 * j + 2];
		const float d = ray_tri_strict(o, dir, mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), mk3(q1.z, q1.w, q2.x));
		const bool take = (orig != src) && (d > 0.0f) && ((d < bd) || (d == bd && orig < bi));
		bd = take ? d : bd;
		bi = take ? orig : bi;
	}
	uint32_t node = 1, trail = 0;
	// Every step either descends one level or retires one pending sibling, so the walk visits each node at most once;
	// the explicit bound is a belt-and-braces exit condition (a wave that never finishes can take the whole GPU down).
	// A little slack on the cull distance: a box may be entered a few ulps after the hit distance of a triangle inside it.
	for (uint32_t steps = 0, max_steps = 4u * B.n_leaves + 64u; steps < max_steps; ++steps) {
		bool descended = false;
		++n_steps;
		if (node < B.first_leaf) {
			const uint32_t c0 = 2 * node, c1 = c0 + 1;
			float t0, t1;
			const float cull = bd * 1.00001f;
			const bool h0 = box_hit(B.nodes[2 * c0], B.nodes[2 * c0 + 1], o, inv, cull, t0);
			const bool h1 = box_hit(B.nodes[2 * c1], B.nodes[2 * c1 + 1], o, inv, cull, t1);
			if (h0 | h1) {
				const bool both = h0 & h1;
				const bool first1 = both ? (t1 < t0) : h1;
				node = first1 ? c1 : c0;
				trail = (trail << 1) | (both ? 1u : 0u);
				descended = true;
			}
		} else {
			const uint32_t leaf = node - B.first_leaf;
			++n_leaves;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t j = leaf * 4 + k;
				const int orig = B.leaf_idx[j];
				const float4 q0 = B.leaf_rec[3 * j], q1 = B.leaf_rec[3 * j + 1], q2 = B.leaf_rec[3 * j + 2];
				const float d = ray_tri_strict(o, dir, mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), mk3(q1.z, q1.w, q2.x));
				const bool take = (orig >= 0) && (orig != src) && (d > 0.0f) && ((d < bd) || (d == bd && orig < bi));
				bd = take ? d : bd;
				bi = take ? orig : bi;
			}
		}
		if (!descended) {
			// pop: climb to the deepest level whose far child is still pending and go to that sibling -- unless the best
			// distance found meanwhile already rules its box out (the box was tested when it was pushed, against an older best)
			bool found = false;
			while (trail != 0) {
				const int up = __builtin_ctz(trail);     // levels to climb
				node >>= up;
				trail >>= up;
				node ^= 1u;                               // the pending (far) sibling
				trail ^= 1u;                              // ... is now taken
				float tn;
				if (box_hit(B.nodes[2 * node], B.nodes[2 * node + 1], o, inv, bd * 1.00001f, tn)) { found = true; break; }
			}
			if (!found) break;
		}
	}
	best_d = bd;
	best_i = bi;
	if (steps_out) { *steps_out = n_steps; *leaves_out = n_leaves; }
}

// kernels: the same integrator / flat / hit bodies as the exact scans (sp_kernels.h), with the BVH as the scan
template <int MODE /* 0 flat, 1 pt, 2 hits */>
__global__ void __launch_bounds__(256) k_accel(const KArgs a, const BvhArgs B, const int* __restrict__ src_idx,
                                               int* __restrict__ out_idx, float* __restrict__ out_d) {
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	const bool valid = k < a.n_rays;
	const uint32_t kk = valid ? k : a.n_rays - 1;
	const float* r = a.rays + (size_t)kk * 6;
	const f3 po = mk3(r[0], r[1], r[2]), pdir = mk3(r[3], r[4], r[5]);
	if (MODE == 2) {
		float bd; int bi;
#ifdef SP_BVH_STATS
		uint32_t st = 0, lv = 0;
		scan_bvh(B, po, pdir, src_idx ? src_idx[kk] : -1, bd, bi, &st, &lv);
		wave_add_scans(a.scans + 1, st); wave_add_scans(a.scans + 2, lv);
#else
		scan_bvh(B, po, pdir, src_idx ? src_idx[kk] : -1, bd, bi);
#endif
		if (valid) { out_idx[k] = bi; out_d[k] = bd; }
		wave_add_scans(a.scans, valid ? 1u : 0u);
		return;
	}
	if (MODE == 0) {
		float bd; int bi;
		scan_bvh(B, po, pdir, -1, bd, bi);
		uint32_t px = 0;
		if (bi >= 0) { const float* m = a.mats + (size_t)bi * 6; px = vec3_rgba(mk3(m[0], m[1], m[2])); }
		if (valid) a.out_rgba[k] = px;
		wave_add_scans(a.scans, valid ? 1u : 0u);
		return;
	}
	const uint32_t pixel = (uint32_t)shard_pixel(a, kk);
	uint32_t my_scans = 0;
	f3 accum = mk3(0.0f, 0.0f, 0.0f);
	for (uint32_t s = 0; s < a.n_samples; ++s) {
		f3 o = po, dir = pdir;
		int src = -1, hidx[5];
		float hcos[5];
		int nh = 0;
		bool alive = valid;
#pragma unroll
		for (int depth = 0; depth < 5; ++depth) {
			if (alive) {
				float bd; int bi;
				scan_bvh(B, o, dir, src, bd, bi);
				my_scans++;
				if (bi >= 0) {
					const float* tn = a.tris + (size_t)bi * 12 + 9;
					f3 n = mk3(tn[0], tn[1], tn[2]);
					if (dot3(n, dir) > 0.0f) n = scale3(n, -1.0f);
					double r1, r2;
					philox_uniforms(a.seed, pixel, s, (uint32_t)depth, &r1, &r2);
					const f3 nd = rand_unit_vec(n, r1, r2);
					hcos[depth] = dot3(nd, n);
					hidx[depth] = bi;
					o = add3(o, scale3(dir, bd));
					dir = nd;
					src = bi;
					nh = depth + 1;
				} else {
					alive = false;
				}
			}
		}
		f3 rec = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
		for (int depth = 4; depth >= 0; --depth) {
			if (depth < nh) {
				const float* m = a.mats + (size_t)hidx[depth] * 6;
				const f3 brdf = scale3(mk3(m[0], m[1], m[2]), kInvPi);
				rec = add3(mk3(m[3], m[4], m[5]), scale3(scale3(mul3(brdf, rec), hcos[depth]), kInvP));
			}
		}
		accum = add3(accum, rec);
	}
	accum = scale3(accum, a.inv_n);
	if (valid) {
		a.out_rgba[k] = vec3_rgba(mk3(clamp01(accum.x), clamp01(accum.y), clamp01(accum.z)));
		if (a.out_accum) { a.out_accum[(size_t)k * 3] = accum.x; a.out_accum[(size_t)k * 3 + 1] = accum.y; a.out_accum[(size_t)k * 3 + 2] = accum.z; }
	}
	wave_add_scans(a.scans, my_scans);
}

} // namespace sp
