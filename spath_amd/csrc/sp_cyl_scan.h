// Two-stage closest-hit scan, second generation ("rpl_cyl"): a 7-instruction conservative reject per (ray, triangle),
// survivors remembered as ONE BIT per (4-triangle group, ray) in a per-lane register word, the reference's exact
// Moeller-Trumbore (ray_tri_strict) for the bits that are set.  Bit-identical results.
//
// Stage 1.  sp_filter_scan.h rejects a pair when the ray misses the slab spanned by the triangle's longest edge and
// the parallel line through the opposite vertex:  |gm| - |t| > Dq  with  gm = w.P - dir.Mc  (6 multiply-adds),
// t = dir.h (3), P = pos x dir, w = unit(longest edge), Mc = w x (mid-point of the two lines' anchors), h = w x
// (their half-difference); DESIGN.md section 4 proves that this never rejects a pair geom::ray_intersect accepts
// (geom.h:197-222).  Here |t| is replaced by an upper bound that costs one multiply-add instead of three,
//     |t| = |dir.h| <= |dir|_2 |h|_2 =: D * H          (H per triangle, D per ray, both rounded up),
//     reject  <=>  |gm| - H*D > Dq,
// geometrically "the ray misses the infinite cylinder of radius (triangle height)/2 around the slab's mid-line".
// Whatever this rejects the slab test rejects too, so conservativeness is inherited (DESIGN.md section 4.2 does the
// rounding bookkeeping).  And w is scaled so that its LARGEST component is exactly 1: that product needs no
// multiplication, the chain starts from the ray-moment component itself -- 5 multiply-adds for gm.  Triangles are
// therefore streamed in three classes (dominant axis x, y, z; k_cyl_scatter), the kernel rotating (P.x,P.y,P.z)
// between classes; the stream order is no longer the index order, which is why stage 2 compares (d, index)
// lexicographically -- the same closest hit and the same tie rule as "ascending index, first strictly smaller d wins"
// (cpu_renderer.cpp:39-49).
//
// Survivor bookkeeping.  About twice as many pairs survive the cylinder test as the slab test (0.5 % against 0.24 % at
// BASELINE configs[2]), and in the first generation it was the bookkeeping, not the arithmetic, that cost: a wave-level
// branch per 8 pairs, taken 70 % of the time, with eight compare-and-push blocks behind it, and an LDS queue per lane.
// Here each lane folds the four x = |gm| - H*D of a group into their minimum, subtracts the margin and shifts the SIGN
// BIT of the difference into a 32-bit word (v_min3_f32, v_min_f32, v_sub_f32, v_alignbit_b32: one instruction per pair,
// no branch, no VCC).  At the end of a tile the set bits are resolved (stage 2): the group's four records are re-read, the four x
// recomputed, ray_tri_strict run on each survivor -- by the lane that owns the ray (scan_cyl: 1-2 rays per lane), or by whichever
// lane of the wave is free (scan_cylw further down: 4 rays per lane, the default).  There is no queue, hence no overflow: a scene
// of huge triangles degrades smoothly to the exact-only scan.
//
// record: 6 floats + the index (class a, with (a,b,c) a cyclic rotation of (x,y,z)):
//   q0 = w_b/w_a  w_c/w_a  Mc.x/w_a  Mc.y/w_a        (Mc.z/w_a, H/|w_a|)        bits(original index)
// A GROUP of four triangles is eight float4 "chunks" (128 B): chunks 0-3 = q0 of triangles 0-3, chunk 4 = (Mz0 H0 Mz1 H1),
// chunk 5 = (Mz2 H2 Mz3 H3), chunk 6 = the four indices, chunk 7 unused -- stage 1 reads six 16-B chunks per group (six
// ds_read_b128, 4 LDS cycles each; with a (q0, q1) pair per triangle it needed the .xy of four q1: two ds_read2st64_b64 at
// 8 cycles each).  Tiles are stored chunk-major (cyl_slot).
#pragma once

#include "sp_kernels.h"
#include "sp_filter_scan.h"

namespace sp {

#ifndef SP_CYL_TILE
#define SP_CYL_TILE 384
#endif
// Triangles per LDS tile.  Stage 2 runs once per tile; with the per-lane form it lasts as long as the lane with the most set bits
// needs, and the maximum over 64 lanes of a count grows more slowly than its mean, so larger tiles mean fewer stage-2 rounds per
// triangle (measured, 4 rays per lane: 12.85 rounds per 256 triangles with 256-triangle tiles, 12.09 with 384); the wave-shared
// form fills its 64-entry rounds better with more entries per slot.  384 x 32 B, double-buffered, plus the bit words of scan_cyl
// (12 KB at 4 rays per lane) or the lists and best-hit cells of scan_cylw (14 KB) is what four workgroups per CU can afford.
constexpr uint32_t kCylTile = SP_CYL_TILE;
static_assert(kCylTile % 128 == 0, "whole waves per LDS-DMA pass, whole words of group bits");
constexpr uint32_t kCylTileQ = 2u * kCylTile;     // float4 per tile
constexpr uint32_t kCylGroups = kCylTile / 4u;    // groups of 4 triangles per tile
// A tile is stored CHUNK-MAJOR, in the global stream and in LDS alike: float4 number c (0..7: records 0..3 of a group, two float4
// each) of group g sits at position c * kCylGroups + g.  Stage 1 reads a group at a wave-uniform g (broadcast reads with immediate
// offsets c * kCylGroups * 16 B); stage 2 reads at a PER-LANE group, and then consecutive groups are consecutive 16-B slots --
// 16 bank phases.  With the records of a group contiguous (g * 128 B + const) every lane of an instruction would fall on one of
// two bank phases: ~32-way conflicts, a third of all LDS cycles (profiles/r02_pmc_per_lane_vs_wave_shared_stage2_spp16.txt).
SP_DEV constexpr uint32_t cyl_slot(uint32_t group, uint32_t chunk) { return chunk * kCylGroups + group; }

struct CylStream {
	const float4* rec;       // class-major stream; every class starts on a tile boundary
	const uint32_t* hdr;     // [0..2] triangles per class, [3..5] first tile of each class (k_cyl_offsets)
	const float4* big;       // sp_cylm_scan.h only: exact records of the big class (3 float4 each), hdr[8] of them
};

// triangle u of group grp of a tile: q0 -> chunk u; (q1.x, q1.y) = (Mz, H) -> chunk 4 + u/2, half u%2; q1.z = index bits -> chunk 6, lane u
SP_DEV void cyl_store(float4* __restrict__ tile, uint32_t grp, uint32_t u, const float4 q0, const float4 q1) {
	tile[cyl_slot(grp, u)] = q0;
	float* mh = (float*)(tile + cyl_slot(grp, 4u + (u >> 1))) + 2u * (u & 1u);
	mh[0] = q1.x; mh[1] = q1.y;
	((float*)(tile + cyl_slot(grp, 6u)))[u] = q1.z;
	if (u == 0u) tile[cyl_slot(grp, 7u)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// the stage-1 data of a group: six float4
struct CylGroup {
	float4 q0[4];
	float4 mh01, mh23;      // (Mz, H) of triangles 0,1 and 2,3
};
SP_DEV CylGroup cyl_group(const float4* tile, uint32_t grp) {
	CylGroup G;
#pragma unroll
	for (int u = 0; u < 4; ++u) G.q0[u] = tile[cyl_slot(grp, (uint32_t)u)];
	G.mh01 = tile[cyl_slot(grp, 4u)]; G.mh23 = tile[cyl_slot(grp, 5u)];
	return G;
}
SP_DEV float cyl_mz(const CylGroup& G, int u) { return u == 0 ? G.mh01.x : u == 1 ? G.mh01.z : u == 2 ? G.mh23.x : G.mh23.z; }
SP_DEV float cyl_h(const CylGroup& G, int u) { return u == 0 ? G.mh01.y : u == 1 ? G.mh01.w : u == 2 ? G.mh23.y : G.mh23.w; }

// ---- record of one triangle (double arithmetic, each coefficient rounded once).  Returns the class.
SP_DEV int cyl_record(const float* __restrict__ t, uint32_t idx, float4& q0, float4& q1) {
	const double A[3] = { t[0], t[1], t[2] }, B[3] = { t[3], t[4], t[5] }, C[3] = { t[6], t[7], t[8] };
	// e1, e2 exactly as the reference rounds them (geom.h:200-201); c = e2 - e1 is the third edge
	const double e1[3] = { (double)(t[3] - t[0]), (double)(t[4] - t[1]), (double)(t[5] - t[2]) };
	const double e2[3] = { (double)(t[6] - t[0]), (double)(t[7] - t[1]), (double)(t[8] - t[2]) };
	const double c[3] = { e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2] };
	const double l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
	const double l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
	const double lc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
	// narrowest slab = along the longest edge (sp_filter_scan.h, k_repack_filter)
	const double* w3; const double* p0; const double* p1; double len2;
	if (l2 >= l1 && l2 >= lc) { w3 = e2; p0 = A; p1 = B; len2 = l2; }
	else if (l1 >= lc)        { w3 = e1; p0 = A; p1 = C; len2 = l1; }
	else                      { w3 = c;  p0 = B; p1 = A; len2 = lc; }
	const double len = sqrt(len2);
	const double w[3] = { w3[0] / len, w3[1] / len, w3[2] / len };
	const double ax = fabs(w[0]), ay = fabs(w[1]), az = fabs(w[2]);
	const int a = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
	const int b = a == 2 ? 0 : a + 1, cc = b == 2 ? 0 : b + 1;
	const double s = 1.0 / w[a];
	const double pc[3] = { 0.5 * (p0[0] + p1[0]), 0.5 * (p0[1] + p1[1]), 0.5 * (p0[2] + p1[2]) };
	const double ph[3] = { 0.5 * (p1[0] - p0[0]), 0.5 * (p1[1] - p0[1]), 0.5 * (p1[2] - p0[2]) };
	const double mc[3] = { (w[1] * pc[2] - w[2] * pc[1]) * s, (w[2] * pc[0] - w[0] * pc[2]) * s, (w[0] * pc[1] - w[1] * pc[0]) * s };
	const double hv[3] = { (w[1] * ph[2] - w[2] * ph[1]) * s, (w[2] * ph[0] - w[0] * ph[2]) * s, (w[0] * ph[1] - w[1] * ph[0]) * s };
	const float beta = (float)(w[b] * s), gamma = (float)(w[cc] * s);
	const float mx = (float)mc[0], my = (float)mc[1], mz = (float)mc[2];
	// H >= |h|_2 after every rounding: (1 + 2^-20) covers the double evaluation and the conversion to float
	float H = (float)(sqrt(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]) * (1.0 + 0x1p-20));
	// anything non-finite (degenerate triangle: len = 0; NaN or inf vertices; overflow of the moments): a record
	// that always survives -- H = +inf makes x = -inf for every ray
	const float chk = beta + gamma + mx + my + mz + H;
	const bool ok = (chk - chk) == 0.0f;
	q0 = ok ? make_float4(beta, gamma, mx, my) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	q1 = make_float4(ok ? mz : 0.0f, ok ? H : __builtin_inff(), __uint_as_float(idx), 0.0f);
	return ok ? a : 0;
}

SP_DEV int cyl_class(const float* __restrict__ t) {
	float4 q0, q1;
	return cyl_record(t, 0u, q0, q1);
}

// ---- pass 1: triangles of each class in every block of 256
__global__ void __launch_bounds__(256) k_cyl_count(const float* __restrict__ tris, uint32_t n, uint32_t* __restrict__ block_counts) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	const int cls = i < n ? cyl_class(tris + (size_t)i * 12) : -1;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const int cnt = __syncthreads_count(cls == k);
		if (threadIdx.x == 0) block_counts[(size_t)blockIdx.x * 3 + k] = (uint32_t)cnt;
	}
}

// ---- pass 2 (one workgroup): exclusive prefix of the block counts per class; hdr[0..2] = class sizes,
// hdr[3..5] = first tile of each class, hdr[6] = tiles in total
__global__ void __launch_bounds__(256) k_cyl_offsets(uint32_t* __restrict__ block_counts, uint32_t nblocks, uint32_t* __restrict__ hdr, uint32_t tile_sz) {
	__shared__ uint32_t part[256];
	__shared__ uint32_t total[3];
	const uint32_t tid = threadIdx.x;
	const uint32_t per = (nblocks + 255u) / 256u;
	const uint32_t lo = tid * per < nblocks ? tid * per : nblocks, hi = lo + per < nblocks ? lo + per : nblocks;
	for (int k = 0; k < 3; ++k) {
		uint32_t sum = 0;
		for (uint32_t b = lo; b < hi; ++b) sum += block_counts[(size_t)b * 3 + k];
		part[tid] = sum;
		__syncthreads();
		if (tid == 0) {
			uint32_t run = 0;
			for (int j = 0; j < 256; ++j) { const uint32_t v = part[j]; part[j] = run; run += v; }
			total[k] = run;
		}
		__syncthreads();
		uint32_t run = part[tid];
		for (uint32_t b = lo; b < hi; ++b) { const uint32_t v = block_counts[(size_t)b * 3 + k]; block_counts[(size_t)b * 3 + k] = run; run += v; }
		__syncthreads();
	}
	if (tid == 0) {
		uint32_t tile = 0;
		for (int k = 0; k < 3; ++k) { hdr[k] = total[k]; hdr[3 + k] = tile; tile += (total[k] + tile_sz - 1u) / tile_sz; }
		hdr[6] = tile;
	}
}

// ---- pass 3: write every triangle's record to its place (stable within a class: ascending original index)
__global__ void __launch_bounds__(256) k_cyl_scatter(const float* __restrict__ tris, uint32_t n, const uint32_t* __restrict__ block_offsets,
                                                    const uint32_t* __restrict__ hdr, float4* __restrict__ rec) {
	__shared__ uint32_t wave_cnt[4][3];
	const uint32_t tid = threadIdx.x, i = blockIdx.x * 256u + tid, wv = tid >> 6, lane = tid & 63u;
	float4 q0, q1;
	q0 = q1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	const int cls = i < n ? cyl_record(tris + (size_t)i * 12, i, q0, q1) : -1;
	uint32_t rank = 0;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const unsigned long long m = __ballot(cls == k);
		if (cls == k) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		if (lane == 0) wave_cnt[wv][k] = (uint32_t)__popcll(m);
	}
	__syncthreads();
	if (cls >= 0) {
		for (uint32_t v = 0; v < wv; ++v) rank += wave_cnt[v][cls];
		const size_t pos = (size_t)hdr[3 + cls] * kCylTile + block_offsets[(size_t)blockIdx.x * 3 + cls] + rank;
		const size_t tile = pos / kCylTile;
		const uint32_t in_tile = (uint32_t)(pos - tile * kCylTile), grp = in_tile >> 2, u = in_tile & 3u;
		cyl_store(rec + tile * kCylTileQ, grp, u, q0, q1);
	}
}

// ---- pass 4: the ragged end of each class's last tile.  H = -inf: x = +inf, rejected by every ray that has a finite
// margin; a ray whose filter is off (margin +inf) sends it to stage 2, where index n_tris is a zero exact record
// (k_repack pads the exact stream), i.e. a = 0, rejected at geom.h:204
__global__ void __launch_bounds__(256) k_cyl_pad(const uint32_t* __restrict__ hdr, uint32_t n_tris, float4* __restrict__ rec) {
	const uint32_t k = blockIdx.x, tid = threadIdx.x;
	const uint32_t n = hdr[k], first = hdr[3 + k] * kCylTile;
	for (uint32_t pos = n + tid; pos < (n + kCylTile - 1u) / kCylTile * kCylTile; pos += 256u) {
		const size_t gpos = (size_t)first + pos, tile = gpos / kCylTile;
		const uint32_t in_tile = (uint32_t)(gpos - tile * kCylTile), grp = in_tile >> 2, u = in_tile & 3u;
		cyl_store(rec + tile * kCylTileQ, grp, u, make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, -__builtin_inff(), __uint_as_float(n_tris), 0.0f));
	}
}

// one tile global -> LDS by LDS-DMA: 16 B per lane, each wave instruction landing 1 KB contiguously
SP_DEV void cyl_tile_dma(const float4* __restrict__ src, float4* dst, uint32_t tid, uint32_t wbase) {
	typedef __attribute__((address_space(1))) const void* gptr_t;
	typedef __attribute__((address_space(3))) void* lptr_t;
	static_assert(kCylTileQ % 256 == 0, "whole workgroup passes");
#pragma unroll
	for (int p = 0; p < (int)(kCylTileQ / 256u); ++p)
		__builtin_amdgcn_global_load_lds((gptr_t)(src + p * 256 + tid), (lptr_t)(dst + p * 256 + wbase), 16, 0, 0);
}

// x = |gm| - H*D for one (record, ray): 6 VALU
SP_DEV float cyl_x(const float4 q0, float mz, float H, float Pa, float Pb, float Pc, float ndx, float ndy, float ndz, float D) {
	float gm = __builtin_fmaf(q0.x, Pb, Pa);
	gm = __builtin_fmaf(q0.y, Pc, gm);
	gm = __builtin_fmaf(ndx, q0.z, gm);
	gm = __builtin_fmaf(ndy, q0.w, gm);
	gm = __builtin_fmaf(ndz, mz, gm);
	return __builtin_fmaf(-H, D, __builtin_fabsf(gm));
}

template <int R>
struct CylRay {            // per-slot filter state
	float Pa[R], Pb[R], Pc[R];     // ray moment pos x dir, rotated to the current class
	float ndx[R], ndy[R], ndz[R];  // -dir
	float D[R], Dq[R];             // |dir|_2 rounded up; margin (sp_filter_scan.h) a hair above, or +-inf
};

// per-slot filter state of the R rays of a lane (shared by scan_cyl and scan_cylw)
template <int R>
SP_DEV void cyl_setup(float rv, const RaySlots<R>& s, CylRay<R>& f) {
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const f3 P = cross3(s.o[r], s.dir[r]);
		const float adx = fabsf(s.dir[r].x), ady = fabsf(s.dir[r].y), adz = fabsf(s.dir[r].z);
		const float dn = adx + ady + adz;
		const float on = fabsf(s.o[r].x) + fabsf(s.o[r].y) + fabsf(s.o[r].z);
		const float dmax = fmaxf(adx, fmaxf(ady, adz));
		// every product of stage 1 is bounded by mag (x sqrt(3) for the scaled w), a sum of six by 11 mag: below 1e37 nothing
		// overflows.  |dir|^2 must neither overflow nor underflow: largest component in [1e-18, 1e18].  Outside (or NaN):
		// the filter is off for this ray -- zero moment, margin +inf: every pair survives, no NaN can arise
		const float mag = dn * (on + 2.0f * rv);
		const bool fin = (mag < 1e37f) && (dmax > 1e-18f) && (dmax < 1e18f);
		const bool on_ = fin && s.act[r];
		f.Pa[r] = on_ ? P.x : 0.0f; f.Pb[r] = on_ ? P.y : 0.0f; f.Pc[r] = on_ ? P.z : 0.0f;
		f.ndx[r] = on_ ? -s.dir[r].x : 0.0f; f.ndy[r] = on_ ? -s.dir[r].y : 0.0f; f.ndz[r] = on_ ? -s.dir[r].z : 0.0f;
		f.D[r] = on_ ? __builtin_sqrtf(s.dir[r].x * s.dir[r].x + s.dir[r].y * s.dir[r].y + s.dir[r].z * s.dir[r].z) * (1.0f + 0x1p-21f) : 1.0f;
		// survive <=> !(x > Dq); tested as sign(x - Dq') with Dq' a hair above Dq (and > 0), so that x == Dq survives too
		const float dq = fmaxf(0x1p-16f * 1.01f * mag * (1.0f + 0x1p-20f), 1e-37f);
		f.Dq[r] = !s.act[r] ? -__builtin_inff() : (fin ? dq : __builtin_inff());     // inactive slot: x - (-inf) = +inf, rejected
	}
}

// Closest hit for the R rays of every lane.  Block-uniform call (barriers inside).
template <int R>
SP_DEV void scan_cyl(const KArgs& a, const CylStream cs, float rv, const RaySlots<R>& s, float (&bd)[R], int (&bi)[R]) {
	__shared__ float4 sm[2 * kCylTileQ];
	static_assert(R == 1 || R == 2 || R == 4, "bits per group must divide 32");
	constexpr uint32_t kGPW = 32u / R;                 // groups of 4 triangles per 32-bit word
	constexpr int kNW = (int)((kCylTile / 4u) / kGPW);   // words per tile
	// the tile's bit words, [word][thread]: written once per word by the hot loop, read back by stage 2 when a lane moves on to
	// its next non-empty word (in registers they would have to be picked and cleared through select cascades)
	__shared__ uint32_t wq[kNW * 256];
	const uint32_t tid = threadIdx.x;
	const uint32_t wbase = tid & ~63u;

	CylRay<R> f;
	cyl_setup<R>(rv, s, f);
#pragma unroll
	for (int r = 0; r < R; ++r) { bd[r] = kMaxDist; bi[r] = -1; }

	// the stream is one run of tiles: class 0's, then class 1's, then class 2's (each class starts on a tile boundary)
	const uint32_t total_tiles = cs.hdr[6];
	uint32_t cls = 0;
	__syncthreads();                                  // readers of the previous scan are done with sm
	cyl_tile_dma(cs.rec, sm, tid, wbase);
	__syncthreads();                                  // (the barrier's fence waits for the DMA: vmcnt(0))
	for (uint32_t gt = 0; gt < total_tiles; ++gt) {
		// entering the next class: rotate the moment so that the class's dominant axis sits in Pa (wave-uniform)
		while (cls < 2u && gt >= cs.hdr[4 + cls]) {
#pragma unroll
			for (int r = 0; r < R; ++r) { const float t0 = f.Pa[r]; f.Pa[r] = f.Pb[r]; f.Pb[r] = f.Pc[r]; f.Pc[r] = t0; }
			++cls;
		}
		const float4* cur = sm + (gt & 1u) * kCylTileQ;
		const uint32_t left = cs.hdr[cls] - (gt - cs.hdr[3 + cls]) * kCylTile;
		const uint32_t ngroups = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((left < kCylTile ? left : kCylTile) + 3u) / 4u));   // a scalar loop bound
		// ---- stage 1: one bit per (group, slot); the first group ends up in the most significant bits
		uint32_t nz = 0;                                  // bit wi: this lane's word wi has a bit set
		uint32_t g = 0;
#pragma unroll
		for (int wi = 0; wi < kNW; ++wi) {
			uint32_t wv = 0;
			const uint32_t gend = ngroups < (uint32_t)(wi + 1) * kGPW ? ngroups : (uint32_t)(wi + 1) * kGPW;
			const uint32_t g0 = g;
			for (; g < gend; ++g) {
				const CylGroup G = cyl_group(cur, g);
#pragma unroll
				for (int r = 0; r < R; ++r) {
					float x[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) x[u] = cyl_x(G.q0[u], cyl_mz(G, u), cyl_h(G, u), f.Pa[r], f.Pb[r], f.Pc[r], f.ndx[r], f.ndy[r], f.ndz[r], f.D[r]);
					const float m = __builtin_fminf(__builtin_fminf(x[0], x[1]), __builtin_fminf(x[2], x[3]));
					wv = __builtin_amdgcn_alignbit(wv, __float_as_uint(m - f.Dq[r]), 31);      // (wv << 1) | sign(m - Dq)
				}
			}
			const uint32_t done = (g - g0) * R;                       // bits appended; left-align (wave-uniform shift)
			const uint32_t wd = done == 0u ? 0u : (wv << (32u - done));
			wq[wi * 256 + tid] = wd;
			nz |= (wd != 0u ? 1u : 0u) << wi;
		}
		// the next tile streams in while the survivors are resolved
		if (gt + 1u < total_tiles) cyl_tile_dma(cs.rec + (size_t)(gt + 1u) * kCylTileQ, sm + ((gt + 1u) & 1u) * kCylTileQ, tid, wbase);
#ifdef SP_FILTER_STATS
		uint32_t st_bits = 0, st_rounds = 0, st_exact = 0;      // experiment build only -> a.scans[1..4]
#pragma unroll
		for (int wi = 0; wi < kNW; ++wi) st_bits += (uint32_t)__builtin_popcount(wq[wi * 256 + tid]);
#endif
		// ---- stage 2: every lane walks its set bits; one exact test per lane and round
		uint32_t sub = 0;                                 // candidates of the current group already done (bit u)
		uint32_t wid = nz ? (uint32_t)__builtin_ctz(nz) : 0u;      // current word of this lane and what is left of it
		uint32_t curw = nz ? wq[wid * 256 + tid] : 0u;
		for (;;) {
			if (!__any(nz != 0u)) break;
#ifdef SP_FILTER_STATS
			++st_rounds;
#endif
			if (nz != 0u) {
				// the lane's following non-empty word, requested now and used at the end of the round if the current one runs out
				// (branch-free on purpose: a conditional reload makes the compiler nest the loop, and lanes then wait for
				// each other at word boundaries)
				const uint32_t nz2 = nz & (nz - 1u);
				const uint32_t wid2 = nz2 ? (uint32_t)__builtin_ctz(nz2) : 0u;
				const uint32_t nextw = wq[wid2 * 256 + tid];
				const uint32_t e = (uint32_t)__builtin_clz(curw);             // first set bit: entry within the word
				const uint32_t grp = wid * kGPW + e / R;
				const int slot = (int)(e % R);
				float Pa = f.Pa[0], Pb = f.Pb[0], Pc = f.Pc[0], ndx = f.ndx[0], ndy = f.ndy[0], ndz = f.ndz[0], D = f.D[0], Dq = f.Dq[0];
				float ox = s.o[0].x, oy = s.o[0].y, oz = s.o[0].z;
				float dx = s.dir[0].x, dy = s.dir[0].y, dz = s.dir[0].z;
				int src = s.src[0];
				float best = bd[0]; int besti = bi[0];
#pragma unroll
				for (int r = 1; r < R; ++r) {
					const bool pick = (slot == r);
					Pa = pick ? f.Pa[r] : Pa; Pb = pick ? f.Pb[r] : Pb; Pc = pick ? f.Pc[r] : Pc;
					ndx = pick ? f.ndx[r] : ndx; ndy = pick ? f.ndy[r] : ndy; ndz = pick ? f.ndz[r] : ndz;
					D = pick ? f.D[r] : D; Dq = pick ? f.Dq[r] : Dq;
					ox = pick ? s.o[r].x : ox; oy = pick ? s.o[r].y : oy; oz = pick ? s.o[r].z : oz;
					dx = pick ? s.dir[r].x : dx; dy = pick ? s.dir[r].y : dy; dz = pick ? s.dir[r].z : dz;
					src = pick ? s.src[r] : src;
					best = pick ? bd[r] : best; besti = pick ? bi[r] : besti;
				}
				// the group's four records (per-lane LDS address) and their x again; the first survivor not yet done
				uint32_t cand = 0; int idx = 0;
				const CylGroup G = cyl_group(cur, grp);
				const float4 gi = cur[cyl_slot(grp, 6u)];
#pragma unroll
				for (int u = 3; u >= 0; --u) {
					const float x = cyl_x(G.q0[u], cyl_mz(G, u), cyl_h(G, u), Pa, Pb, Pc, ndx, ndy, ndz, D);
					const bool sv = !(x - Dq >= 0.0f) && !((sub >> u) & 1u);      // the sign-bit decision again; NaN -> survivor
					cand = sv ? (cand | (1u << u)) : cand;
					idx = sv ? (int)__float_as_uint(u == 0 ? gi.x : u == 1 ? gi.y : u == 2 ? gi.z : gi.w) : idx;   // ends as the lowest surviving u's index
				}
				const uint32_t lowest = cand & (0u - cand);
				const bool last = (cand == lowest);                             // no further survivor in this group
				sub = last ? 0u : (sub | lowest);
				curw = last ? (curw & ~(0x80000000u >> e)) : curw;
				if (cand != 0u) {
#ifdef SP_FILTER_STATS
					++st_exact;
#endif
					const float4 x0 = a.scan[3 * (size_t)idx + 0], x1 = a.scan[3 * (size_t)idx + 1], x2 = a.scan[3 * (size_t)idx + 2];
					const float d = ray_tri_strict(mk3(ox, oy, oz), mk3(dx, dy, dz), mk3(x0.x, x0.y, x0.z), mk3(x0.w, x1.x, x1.y), mk3(x1.z, x1.w, x2.x));
					// ascending-index scan with "first strictly smaller d wins" (cpu_renderer.cpp:44) == lexicographic (d, index) minimum
					const bool take = (d > 0.0f) && (idx != src) && ((d < best) || (d == best && idx < besti));
					best = take ? d : best;
					besti = take ? idx : besti;
#pragma unroll
					for (int r = 0; r < R; ++r) { const bool pick = (slot == r); bd[r] = pick ? best : bd[r]; bi[r] = pick ? besti : bi[r]; }
				}
				const bool adv = (curw == 0u);                                    // this word is done: on to the lane's next non-empty one
				nz = adv ? nz2 : nz;
				wid = adv ? wid2 : wid;
				curw = adv ? nextw : curw;
			}
		}
#ifdef SP_FILTER_STATS
		{
			uint32_t v = st_bits, x = st_exact;
			for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); x += __shfl_xor(x, off, 64); }
			if ((tid & 63u) == 0) { atomicAdd(a.scans + 1, (unsigned long long)v); atomicAdd(a.scans + 2, (unsigned long long)st_rounds); atomicAdd(a.scans + 3, 1ull); atomicAdd(a.scans + 4, (unsigned long long)x); }
		}
#endif
		__syncthreads();                        // next tile landed (vmcnt(0) in the fence) and this one is free again
	}
}


// ---------------------------------------------------------------------------------------------------------------------
// scan_cylw: the same stage 1, stage 2 shared by the whole WAVE.
//
// In scan_cyl a lane resolves only its own set bits, so a tile's stage 2 lasts as long as the busiest lane needs: 18 rounds per
// 384-triangle tile with 42 % of the lanes working (profiles/filter_stats.json).  Here the set bits of one slot (ray r of every
// lane) are written out as a list of (lane, group) entries in LDS, and ALL 64 lanes take entries from that list, 64 per round:
//   * the bit words are kept per slot (a group's R sign bits go to R different words: same one instruction per pair), so a
//     slot's bits need no de-interleaving;
//   * the entry's ray is the donor lane's slot-r registers, a FIXED register set for the whole round: twelve ds_bpermute_b32;
//   * the result goes back through one 64-bit LDS atomicMin per hit on the key (float bits of d) << 32 | index -- d > 0, so keys
//     order like (d, index) pairs: the closest hit with the lowest index on ties, i.e. the reference's "ascending index, first
//     strictly smaller d wins" (cpu_renderer.cpp:39-49) in any processing order.  The running best of every ray lives in that
//     cell for the whole scan.
// A wave lists at most kCylCap entries per pass (16 bits each); what does not fit waits in the words for the next pass (rare).
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kCylCap = 64u * 12u;      // list entries per wave and pass (16 bits each): 12 per lane on average

template <int R>
SP_DEV void scan_cylw(const KArgs& a, const CylStream cs, float rv, const RaySlots<R>& s, float (&bd)[R], int (&bi)[R]) {
	__shared__ float4 sm[2 * kCylTileQ];
	__shared__ unsigned short lst[4 * kCylCap];               // per wave: (donor lane << 7) | group within the tile
	__shared__ uint32_t lcnt[4];                              // per wave: entries listed in the current pass
	static_assert(kCylTile / 4u <= 128u, "group index must fit 7 bits");
	__shared__ unsigned long long cell[R * 256];              // per (slot, thread): best (d bits << 32 | index) so far
	constexpr int kW = (int)(kCylTile / 128u);                // 32-bit words per slot and tile (one bit per group of 4 triangles)
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wbase = tid & ~63u;
	unsigned short* const mylst = lst + (tid >> 6) * kCylCap;
	uint32_t* const mycnt = lcnt + (tid >> 6);

	CylRay<R> f;
	cyl_setup<R>(rv, s, f);
	const unsigned long long kNone = ((unsigned long long)__float_as_uint(kMaxDist) << 32) | 0xffffffffull;
#pragma unroll
	for (int r = 0; r < R; ++r) cell[r * 256 + tid] = kNone;

	const uint32_t total_tiles = cs.hdr[6];
	uint32_t cls = 0;
	__syncthreads();                                  // readers of the previous scan are done with sm
	cyl_tile_dma(cs.rec, sm, tid, wbase);
	__syncthreads();                                  // (the barrier's fence waits for the DMA: vmcnt(0))
	for (uint32_t gt = 0; gt < total_tiles; ++gt) {
		while (cls < 2u && gt >= cs.hdr[4 + cls]) {
#pragma unroll
			for (int r = 0; r < R; ++r) { const float t0 = f.Pa[r]; f.Pa[r] = f.Pb[r]; f.Pb[r] = f.Pc[r]; f.Pc[r] = t0; }
			++cls;
		}
		const float4* cur = sm + (gt & 1u) * kCylTileQ;
		const uint32_t left = cs.hdr[cls] - (gt - cs.hdr[3 + cls]) * kCylTile;
		const uint32_t ngroups = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((left < kCylTile ? left : kCylTile) + 3u) / 4u));   // a scalar loop bound
		// ---- stage 1: word[r][wi] bit (31 - k) = group 32*wi + k of this tile survives for ray r
		uint32_t word[R][kW];
		uint32_t g = 0;
#pragma unroll
		for (int wi = 0; wi < kW; ++wi) {
			uint32_t wv[R];
#pragma unroll
			for (int r = 0; r < R; ++r) wv[r] = 0;
			const uint32_t gend = ngroups < (uint32_t)(wi + 1) * 32u ? ngroups : (uint32_t)(wi + 1) * 32u;
			const uint32_t g0 = g;
			for (; g < gend; ++g) {
				const CylGroup G = cyl_group(cur, g);
#pragma unroll
				for (int r = 0; r < R; ++r) {
					float x[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) x[u] = cyl_x(G.q0[u], cyl_mz(G, u), cyl_h(G, u), f.Pa[r], f.Pb[r], f.Pc[r], f.ndx[r], f.ndy[r], f.ndz[r], f.D[r]);
					const float m = __builtin_fminf(__builtin_fminf(x[0], x[1]), __builtin_fminf(x[2], x[3]));
					wv[r] = __builtin_amdgcn_alignbit(wv[r], __float_as_uint(m - f.Dq[r]), 31);      // (wv << 1) | sign(m - Dq)
				}
			}
			const uint32_t done = g - g0;                             // bits appended; left-align (wave-uniform shift)
#pragma unroll
			for (int r = 0; r < R; ++r) word[r][wi] = done == 0u ? 0u : (wv[r] << (32u - done));
		}
		// the next tile streams in while the survivors are resolved
		if (gt + 1u < total_tiles) cyl_tile_dma(cs.rec + (size_t)(gt + 1u) * kCylTileQ, sm + ((gt + 1u) & 1u) * kCylTileQ, tid, wbase);
		// ---- stage 2, slot by slot
#pragma unroll
		for (int r = 0; r < R; ++r) {
			for (;;) {                                                // passes: one, unless the wave's list overflows
				uint32_t c = 0;
#pragma unroll
				for (int wi = 0; wi < kW; ++wi) c += (uint32_t)__builtin_popcount(word[r][wi]);
				if (!__any(c != 0u)) break;
				// list space: one LDS atomic per lane (any disjoint allocation will do: the results meet in an order-independent atomicMin)
				if (lane == 0) *mycnt = 0u;
				uint32_t j = c ? atomicAdd(mycnt, c) : 0u;
				const uint32_t jend = j + c < kCylCap ? j + c : kCylCap;     // what does not fit stays in the words for the next pass
#pragma unroll
				for (int wi = 0; wi < kW; ++wi) {
					uint32_t m = word[r][wi];
					while (__any(m != 0u && j < jend)) {
						if (m != 0u && j < jend) {
							const uint32_t e = (uint32_t)__builtin_clz(m);
							mylst[j++] = (unsigned short)((lane << 7) | ((uint32_t)wi * 32u + e));
							m &= ~(0x80000000u >> e);
						}
					}
					word[r][wi] = m;
				}
				uint32_t total = *mycnt;
				total = (uint32_t)__builtin_amdgcn_readfirstlane((int)(total < kCylCap ? total : kCylCap));
#ifdef SP_FILTER_STATS
				if (lane == 0) { atomicAdd(a.scans + 1, (unsigned long long)total); atomicAdd(a.scans + 2, (unsigned long long)((total + 63u) / 64u)); }
#endif
				// rounds: 64 entries at a time, one per lane
				for (uint32_t base = 0; base < total; base += 64u) {
					const uint32_t ent = base + lane;
					const bool ok = ent < total;
					const uint32_t entry = ok ? (uint32_t)mylst[ent] : (lane << 7);       // idle lanes: their own ray, group 0, result discarded
					const int L = (int)(entry >> 7);
					const uint32_t grp = entry & 127u;
					// the donor's ray r: a fixed register set, fetched across lanes (every lane takes part in the permutes)
					const float ox = __shfl(s.o[r].x, L, 64), oy = __shfl(s.o[r].y, L, 64), oz = __shfl(s.o[r].z, L, 64);
					const float dx = __shfl(s.dir[r].x, L, 64), dy = __shfl(s.dir[r].y, L, 64), dz = __shfl(s.dir[r].z, L, 64);
					const int src = __shfl(s.src[r], L, 64);
					const float Pa = __shfl(f.Pa[r], L, 64), Pb = __shfl(f.Pb[r], L, 64), Pc = __shfl(f.Pc[r], L, 64);
					const float D = __shfl(f.D[r], L, 64), Dq = __shfl(f.Dq[r], L, 64);
					// (a ray whose filter is off has a zero moment and Dq = +inf: it survives whatever x comes out, so -dir serves for all)
					// the group's four records and their x again: which of the four survive
					uint32_t cand = 0;
					const CylGroup G = cyl_group(cur, grp);
					const float4 gi = cur[cyl_slot(grp, 6u)];
					const int idx4[4] = { (int)__float_as_uint(gi.x), (int)__float_as_uint(gi.y), (int)__float_as_uint(gi.z), (int)__float_as_uint(gi.w) };
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const float x = cyl_x(G.q0[u], cyl_mz(G, u), cyl_h(G, u), Pa, Pb, Pc, -dx, -dy, -dz, D);
						cand |= (ok && !(x - Dq >= 0.0f)) ? (1u << u) : 0u;          // the sign-bit decision again; NaN -> survivor
					}
					// one exact test per lane and turn: 1.09 per entry on average, 1.57 wave-wide turns per round (some lane has two or three)
					while (__any(cand != 0u)) {
#ifdef SP_FILTER_STATS
						if (lane == 0) atomicAdd(a.scans + 5, 1ull);
#endif
						if (cand != 0u) {
							const uint32_t low = cand & (0u - cand);
							cand ^= low;
							const int idx = (low & 1u) ? idx4[0] : (low & 2u) ? idx4[1] : (low & 4u) ? idx4[2] : idx4[3];
#ifdef SP_FILTER_STATS
							atomicAdd(a.scans + 4, 1ull);
#endif
							const float4 x0 = a.scan[3 * (size_t)idx + 0], x1 = a.scan[3 * (size_t)idx + 1], x2 = a.scan[3 * (size_t)idx + 2];
							const float d = ray_tri_strict(mk3(ox, oy, oz), mk3(dx, dy, dz), mk3(x0.x, x0.y, x0.z), mk3(x0.w, x1.x, x1.y), mk3(x1.z, x1.w, x2.x));
							// cpu_renderer.cpp:44: cur_d > 0 && cur_d < d, d starting at MAX_VALUE_DIST; ties -> lowest index: the key's low word
							if ((d > 0.0f) && (d < kMaxDist) && (idx != src))
								atomicMin(&cell[r * 256 + (int)wbase + L], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(uint32_t)idx);
						}
					}
				}
			}
		}
#ifdef SP_FILTER_STATS
		if (lane == 0) atomicAdd(a.scans + 3, 1ull);
#endif
		__syncthreads();                        // next tile landed (vmcnt(0) in the fence) and this one is free again
	}
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const unsigned long long k = cell[r * 256 + tid];
		bd[r] = __uint_as_float((uint32_t)(k >> 32));
		bi[r] = (int)(uint32_t)k;                          // 0xffffffff = -1: no hit
	}
}

} // namespace sp
