// Third generation of the closest-hit scan: the cylinder test of sp_cyl_scan.h with its side product on the FP16 MATRIX pipe.
// Same closest hit, same ties as cpu_renderer.cpp:36-49 (stage 2 is scan_cylw's: exact test of every survivor, best (d, index)).
//
// Stage 1 of sp_cyl_scan.h spends 5 of its 7 VALU instructions per (ray, triangle) on gm = P_a + b P_b + c P_c - dir.Mc'
//  -- a bilinear form of a 5-vector of the ray and a 5-vector of the triangle, i.e. a matrix product over all
// pairs.  The f32 matrix instructions run at the VALU's rate; the f16 ones at 16x.  gm needs ~2^-19 relative accuracy
// (the margin Dq is 2^-16 of the magnitudes involved), which one half-precision product does not give, but three do:
//     v = hi + lo  (hi = half(v), lo = half(v - hi);  |v - hi - lo| <= 2^-22 |v|)
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi       (each product of two halves is exact in the f32 accumulator)
// The five products take 15 of the K = 16 slots of ONE v_mfma_f32_32x32x16_f16 and P_a (coefficient 1) the sixteenth -- as a
// single half: the test is invariant under scaling the ray, so the ray is scaled by t = half(P_a)/P_a (|t - 1| <= 2^-11), which
// makes its P_a exactly a half (when P_a is too small for that, t = 1 and the dropped part, < 2^-22, is added to the margin).
// One instruction = 32 triangles (A rows) x 32 rays (B columns) = 1024 side products in 32 cycles of the matrix
// pipe.  What is left for the VALU is the cylinder bound and the sign bit, ONE bound per group of four triangles:
//     min_i (|gm_i| - H_i D)  >=  min_i |gm_i| - Hmax D      (Hmax = the largest H of the group, D > 0)
// so  m = min(min3(|g0|, |g1|, |g2|), |g3|),  x = fma(-Hmax, D, m),  sign(x - Dq)  -- 5 instructions per 4 pairs instead of the
// 8 that four separate x cost (round 2).  It can only keep MORE than the per-triangle form (every triangle of a rejected group
// has |gm_i| - H_i D > Dq, the condition DESIGN.md 4.2 proves safe), and stage 2 re-tests each triangle of a surviving group with
// its own H.  The prepass orders every class by H (k_cylm_keys + sp_radix_sort.h), so Hmax ~ H_i and next to nothing is lost.
// And the margin is FOLDED INTO THE BOUND, once per tile and ray instead of once per group (k_cylm_hmax, cylm_tile_bound):
//     D' = D + kappa Dq  with  Hmax kappa >= 1 for every group of the tile   =>   Hmax D' >= Hmax D + Dq,
// so  x' = fma(-Hmax, D', m),  sign(x')  -- 4 instructions per 4 pairs -- rejects only what the subtracted form rejects, and x',
// rounded once, has the sign of the exact value.  Stage 1 is where the vector ALU is the limit: the instruction is worth 4 - 5 %.
//
// Scaling.  Halves hold 6e-5 .. 65504, so both sides are scaled by EXACT powers of two: S = 2^k in [2Rv, 4Rv) for lengths
// (Rv = the scene bound of sp_filter_scan.h), s_d = 2^-e for the ray direction (largest |dir| component in [1, 2) afterwards):
//     triangle side:  16 b, 16 c, 16 Mc'/S          (|.| <= 16 and <= 14)         H^ = 256 H / S
//     ray side:       16 s_d P/S, 16 s_d (-dir)     (|.| <= 2^15 and <= 32)       D^ = s_d D,  Dq^ = 256 s_d Dq / S      (all times t)
// so that every term of gm^ is 256 s_d / S times the term of gm: the test |gm^| - H^ D^ > Dq^ is the test |gm| - H D > Dq.  A ray
// further than 512 S from the origin, or outside the guards of cyl_setup, has its filter off (everything survives) as before.
//
// Error budget (u = 2^-24, X = |dir|(|pos| + 2Rv) sqrt(3) bounding the sum of the |terms| as in DESIGN.md 4.2): splitting
// 3 x 2^-22 = 12u per product, the accumulation of 16 terms in f32 <= 16u (measured on the device: 2^-21.5 = 6u,
// tests/test_hip_device_math.py), the scaling by t 2u, against the 7u of the VALU chain they replace; needed by the proof 152u + (30u - 7u) sqrt(3) = 192u of the 256u the margin provides.
// Absolute floors (subnormal lo parts, 2^-25 per value) are 2^-13 of that after the scaling by 16.
//
// Work shape: ONE ray per lane (the matrix instruction reads a whole 32-triangle fragment with one 16-B LDS read per lane, so
// nothing is gained from several rays per lane).  A wave owns 64 rays = two column blocks; lane l holds, of ray block rb, column
// l & 31 -- its own ray when (l >> 5) == rb, its partner's otherwise -- and the rows (triangles) 8j + 4(l >> 5) + i of a
// fragment: the four i of a j are one GROUP of sp_cyl_scan.h, so the sign bits are per (ray, group) exactly as there.
// Tile: 256 triangles = 16 KB: [8 chunks][64 groups] float4 as in sp_cyl_scan.h -- except chunk 7, whose first 16 slots hold the
// scaled Hmax of the groups, [fragment][lane half] x (the four groups j = 0..3 that half owns): one 16-B read per lane and
// fragment -- followed by the A fragments [8 blocks of 32][64 lanes] x 16 B; 8 fragments x 4 group bits = one 32-bit word per
// ray block (128 / 192-triangle tiles: -15 % / -4 %).  Build hooks for experiments: SP_EXP_NO_STAGE2 (timing only), SP_DBG_ALLBITS, SP_DBG_PRINT, SP_CYLM_UNPINNED.
//
// THIS FILE IS INCLUDED TWICE (sp_cylm_both.h): once per workgroup shape, each copy in a namespace of its own --
//   sp::cylm256   256 threads, 256-triangle tiles (four workgroups per CU), one survivor bit per group of four     scenes below kMBigSceneTris triangles
//   sp::cylm512   512 threads, 512-triangle tiles (two workgroups per CU), one survivor bit per OCTET (kMGrp = 8)   larger scenes
// Larger tiles halve what a tile costs beyond its pairs (LDS-DMA issue, barrier, list pass, the half-filled last round of stage 2):
// +7 % at 10^5 and +8 % at 10^6 triangles; eight waves per barrier wait longer for their slowest member, which costs more than that
// where stage 2 is heavy: -2 % at 10^4 triangles, -20 % on the large-triangle scene (profiles/r03_cyl_scan_experiments.log).
#ifndef SP_CYLM_NS
#error "include sp_cylm_both.h, not this file"
#endif

namespace sp { namespace SP_CYLM_NS {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

// (time = a + b / tile with b / 256 = 19 % of the configs[2] frame for 256-thread workgroups: tiles of 128 / 192 / 256 triangles
// 159.2 / 142.5 / 134.1 ms; a workgroup of 256 cannot afford more than 2 x 16 KB of tile buffers next to its lists)
#ifndef SP_CYLM_TILE
#define SP_CYLM_TILE SP_CYLM_THREADS
#endif

constexpr uint32_t kMThreads = SP_CYLM_THREADS;   // threads per workgroup of the kernels over this scan
constexpr uint32_t kMWaves = kMThreads / 64u;
constexpr uint32_t kMTile = SP_CYLM_TILE;        // triangles per tile
constexpr uint32_t kMGroups = kMTile / 4u;        // 128 groups of four
constexpr uint32_t kMBlocks = kMTile / 32u;       // 16 fragments
// Stage 1 keeps ONE bit per (ray, kMGrp triangles): per QUAD (the group of four, 8 tb + 2 j + hh) or per OCTET (two quads: 8 tb + 4 q + hh and
// that + 2, q = 0, 1).  Octets: 6 instead of 8 vector instructions per eight pairs in stage 1, but eight re-tests per list entry in stage 2 --
// measured: +4 % / +8 % at 10^5 / 10^6 triangles, -8 % at 10^4 (more entries per tile there): quads for the 256-thread shape, octets for the 512-thread one.
#ifndef SP_CYLM_GROUP
#define SP_CYLM_GROUP 4
#endif
constexpr uint32_t kMGrp = SP_CYLM_GROUP;
static_assert(kMGrp == 4u || kMGrp == 8u, "quads or octets (all sixteen rows of a lane as one group: -8 % / -4 % at 10^5 / 10^6 triangles against octets)");
constexpr uint32_t kMBitsFrag = 16u / kMGrp;                 // bits per fragment and lane half
constexpr uint32_t kMFragsPerWord = 32u / kMBitsFrag;
constexpr uint32_t kMWords = (kMBlocks * kMBitsFrag + 31u) / 32u;     // 32-bit words of survivor bits per ray block and tile
constexpr uint32_t kMRecQ = kMGroups * 8u;        // 1024 float4: the f32 part
constexpr uint32_t kMTileQ = kMRecQ + kMBlocks * 64u;   // 2048 float4 = 32 KB
static_assert(kMThreads % 64u == 0u && kMThreads <= 1024u, "whole waves");
static_assert(kMTileQ % kMThreads == 0u, "whole workgroup LDS-DMA passes");
static_assert(kMGroups <= 128u, "7-bit group index in a list entry");
static_assert(kMBlocks * 2u <= kMGroups, "the Hmax table fits chunk 7");
SP_DEV constexpr uint32_t cylm_slot(uint32_t group, uint32_t chunk) { return chunk * kMGroups + group; }
// scaled Hmax of the four groups 8 tb + 2 j + hh (j = 0..3) that lane half hh owns in fragment tb: component j of this float4
SP_DEV constexpr uint32_t cylm_hmq(uint32_t tb, uint32_t hh) { return 7u * kMGroups + tb * 2u + hh; }
// behind that table: .x = kappa of the tile, 1 / (the smallest regular Hmax^ of the tile) rounded up (k_cylm_hmax; cylm_tile_bound)
SP_DEV constexpr uint32_t cylm_kq() { return 7u * kMGroups + kMBlocks * 2u; }
static_assert(kMBlocks * 2u + 1u <= kMGroups, "Hmax table and kappa fit chunk 7");

// length scale of the scene: the power of two in [2 Rv, 4 Rv); 0 = matrix filter off for this scene (Rv outside [1e-30, 1e30])
SP_DEV float cylm_scale(float rv) {
	if (!(rv > 1e-30f && rv < 1e30f)) return 0.0f;
	const uint32_t e = (__float_as_uint(rv) >> 23) & 255u;        // rv in [2^(e-127), 2^(e-126))
	return __uint_as_float((e + 2u) << 23);
}

// float -> half, ONCE: the value that goes into a fragment must be the very value its remainder (or the scale t) was computed
// from.  Without the barrier the compiler converts again where the fragment is assembled (v_cvt_pk_f16_f32 next to the
// v_cvt_f16_f32 used for the remainder), and the two instructions do not round a near-tie the same way: hi and lo then
// belong to different splittings and the product is off by an ulp of the half (found by tools/soak.py on aimed rays).
SP_DEV _Float16 to_half(float v) {
	uint32_t b = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)v);
#ifndef SP_CYLM_UNPINNED                                   // (test builds only: shows that tests/test_hip_robustness.py catches the defect)
	__asm__ volatile("" : "+v"(b));
#endif
	return __builtin_bit_cast(_Float16, (unsigned short)b);
}

SP_DEV void half_split(float v, _Float16& hi, _Float16& lo) {
	hi = to_half(v);
	lo = to_half(v - (float)hi);
}

// triangle `in_tile` of a tile: f32 part as cyl_store (chunks 0-6), the A-fragment row
SP_DEV void cylm_store(float4* __restrict__ tile, uint32_t in_tile, const float4 q0, const float4 q1, float S) {
	const uint32_t grp = in_tile >> 2, u = in_tile & 3u;
	tile[cylm_slot(grp, u)] = q0;
	float* mh = (float*)(tile + cylm_slot(grp, 4u + (u >> 1))) + 2u * (u & 1u);
	mh[0] = q1.x; mh[1] = q1.y;
	((float*)(tile + cylm_slot(grp, 6u)))[u] = q1.z;
	const float m = S > 0.0f ? 16.0f / S : 0.0f;                                        // exact: S is a power of two
	_Float16 bh, bl, ch, cl, xh, xl, yh, yl, zh, zl;
	half_split(S > 0.0f ? 16.0f * q0.x : 0.0f, bh, bl); half_split(S > 0.0f ? 16.0f * q0.y : 0.0f, ch, cl);
	half_split(q0.z * m, xh, xl); half_split(q0.w * m, yh, yl); half_split(q1.x * m, zh, zl);
	const _Float16 z = (_Float16)0.0f;
	const half8 k0 = { bh, bl, bh, ch, cl, ch, xh, xl }, k1 = { xh, yh, yl, yh, zh, zl, zh, S > 0.0f ? (_Float16)16.0f : z };
	half8* frag = (half8*)(tile + kMRecQ) + (in_tile >> 5) * 64u + (in_tile & 31u);      // lane = row (h = 0), row + 32 (h = 1)
	frag[0] = k0; frag[32] = k1;
}

// ---- The "big" class.  A triangle whose cylinder is a sizeable fraction of the scene (walls, floors, ground planes) survives stage 1
// for practically every ray: the filter, its list entry, the f32 re-test and the gather of its exact record are pure overhead on it.
// Such triangles -- at most kMBig of them, cylinder radius H >= S / 64 (S = the scene's length scale), the threshold rising in steps
// of 4 while more than kMBig qualify, none at all if even H >= S does not get below that -- are taken out of the stream: every lane
// runs the reference's test on them for its own ray at the start of a scan, the records arriving through the scalar cache
// (cpu_renderer.cpp:39-49 visits them like any other triangle; the order is immaterial, sp_cyl_scan.h).
constexpr uint32_t kMBig = 64u;
constexpr int kMBigLevels = 4;                      // H / S >= 2^-6, 2^-4, 2^-2, 2^0

SP_DEV int cylm_big_level(float H, float S) {       // the highest level the triangle reaches, -1: none (or a degenerate record: H = +inf stays in the stream)
	if (!(S > 0.0f) || !(H < __builtin_inff())) return -1;
	const float r = H / S;
	return r >= 1.0f ? 3 : r >= 0.25f ? 2 : r >= 0.0625f ? 1 : r >= 0.015625f ? 0 : -1;
}

// ---- prepass 0: how many triangles reach each level -> hdr[9 .. 12] (zeroed by the host)
__global__ void __launch_bounds__(256) k_cylm_count_big(const float* __restrict__ tris, uint32_t n, const unsigned int* __restrict__ bounds, uint32_t* __restrict__ hdr) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	const float S = cylm_scale(__uint_as_float(bounds[0]));
	int lvl = -1;
	if (i < n) {
		float4 q0, q1;
		(void)cyl_record(tris + (size_t)i * 12, i, q0, q1);
		lvl = cylm_big_level(q1.y, S);
	}
#pragma unroll
	for (int k = 0; k < kMBigLevels; ++k) {
		const int cnt = __syncthreads_count(lvl >= k);
		if (threadIdx.x == 0 && cnt) atomicAdd(hdr + 9 + k, (uint32_t)cnt);
	}
}

// the level from which on triangles are "big" in this scene: the lowest one that at most kMBig triangles reach (kMBigLevels: none)
SP_DEV int cylm_big_threshold(const uint32_t* __restrict__ hdr) {
	for (int k = 0; k < kMBigLevels; ++k) if (hdr[9 + k] <= kMBig) return k;
	return kMBigLevels;
}

// ---- prepass 1: sort key of every triangle = class (2 bits; 3 = big) | float bits of H >> 2 (H >= 0 or +inf: the bits order like the
// values), value = the triangle's index; triangles per class -> hdr[0..2], big ones -> hdr[8] (zeroed by the host)
__global__ void __launch_bounds__(256) k_cylm_keys(const float* __restrict__ tris, uint32_t n, const unsigned int* __restrict__ bounds, uint32_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals, uint32_t* __restrict__ hdr) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	const float S = cylm_scale(__uint_as_float(bounds[0]));
	const int thr = cylm_big_threshold(hdr);
	int cls = -1;
	if (i < n) {
		float4 q0, q1;
		cls = cyl_record(tris + (size_t)i * 12, i, q0, q1);
		const int lvl = cylm_big_level(q1.y, S);
		if (lvl >= 0 && lvl >= thr) cls = 3;
		keys[i] = ((uint32_t)cls << 30) | (__float_as_uint(q1.y) >> 2);
		vals[i] = i;
	}
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int cnt = __syncthreads_count(cls == k);
		if (threadIdx.x == 0 && cnt) atomicAdd(hdr + (k < 3 ? k : 8), (uint32_t)cnt);
	}
}

// ---- prepass 2 (one thread): hdr[3..5] = first tile of each class, hdr[6] = tiles in total, hdr[7] = the length scale S
__global__ void k_cylm_hdr(uint32_t* __restrict__ hdr, const unsigned int* __restrict__ bounds) {
	uint32_t tile = 0;
	for (int k = 0; k < 3; ++k) { hdr[3 + k] = tile; tile += (hdr[k] + kMTile - 1u) / kMTile; }
	hdr[6] = tile;
	((float*)hdr)[7] = cylm_scale(__uint_as_float(bounds[0]));
}

// ---- prepass 3: the triangle at sorted position p (classes in order, ascending H within a class; equal keys in index order:
// the sort is stable) goes to rank p - (start of its class) of the class's run of tiles
// The big class (the last hdr[8] places of the sorted order) goes to `big`: the exact record of k_repack (v0, e1, e2) with the
// triangle's index in the unused second word of its third float4.
__global__ void __launch_bounds__(256) k_cylm_scatter(const float* __restrict__ tris, uint32_t n, const uint32_t* __restrict__ sorted,
                                                     const uint32_t* __restrict__ hdr, float4* __restrict__ rec, const float4* __restrict__ scan, float4* __restrict__ big) {
	const uint32_t p = blockIdx.x * 256u + threadIdx.x;
	if (p >= n) return;
	const uint32_t i = sorted[p];
	const float S = ((const float*)hdr)[7];
	const uint32_t n_stream = hdr[0] + hdr[1] + hdr[2];
	if (p >= n_stream) {
		const uint32_t b = p - n_stream;
		const float4 x2 = scan[3 * (size_t)i + 2];
		big[3 * b + 0] = scan[3 * (size_t)i + 0]; big[3 * b + 1] = scan[3 * (size_t)i + 1];
		big[3 * b + 2] = make_float4(x2.x, __uint_as_float(i), 0.0f, 0.0f);
		return;
	}
	float4 q0, q1;
	const int cls = cyl_record(tris + (size_t)i * 12, i, q0, q1);
	const uint32_t c0 = cls == 0 ? 0u : cls == 1 ? hdr[0] : hdr[0] + hdr[1];
	const size_t pos = (size_t)hdr[3 + cls] * kMTile + (p - c0);
	const size_t tile = pos / kMTile;
	cylm_store(rec + tile * kMTileQ, (uint32_t)(pos - tile * kMTile), q0, q1, S);
}

// ---- prepass 4: the ragged end of each class's last tile.  f32 record with H = -inf (stage 2's re-test rejects it; a ray whose
// filter is off sends it on to the exact test, where index n_tris is a zero exact record: a = 0, rejected at geom.h:204) and a
// zero A row
__global__ void __launch_bounds__(256) k_cylm_pad(const uint32_t* __restrict__ hdr, uint32_t n_tris, float4* __restrict__ rec) {
	const uint32_t k = blockIdx.x, tid = threadIdx.x;
	const uint32_t n = hdr[k], first = hdr[3 + k] * kMTile;
	const float S = ((const float*)hdr)[7];
	for (uint32_t pos = n + tid; pos < (n + kMTile - 1u) / kMTile * kMTile; pos += 256u) {
		const size_t gpos = (size_t)first + pos, tile = gpos / kMTile;
		cylm_store(rec + tile * kMTileQ, (uint32_t)(gpos - tile * kMTile), make_float4(0.0f, 0.0f, 0.0f, 0.0f),
		           make_float4(0.0f, -__builtin_inff(), __uint_as_float(n_tris), 0.0f), S);
	}
}

// ---- prepass 5: Hmax^ = 256 max(H) / S of every group (one thread per group, one block per tile): +inf if the group holds a
// degenerate triangle (always survives), -inf if it holds nothing but padding (never does) -- and kappa of the tile.
// Stage 1 tests  min_i |g_i| - Hmax^ D' < 0  with  D' = D^ + kappa Dq^  (cylm_tile_bound): the margin Dq^ folded into the cylinder
// bound, one instruction less per group.  It needs  Hmax^ kappa >= 1  for every regular (finite) group of the tile: then
// Hmax^ D' >= Hmax^ D^ + Dq^, i.e. whatever this form rejects, the form with the margin subtracted rejects too.  kappa = 1 / (smallest
// regular Hmax^).  The classes are sorted by H, so Hmax^ hardly varies inside a tile and next to nothing is lost; where it does vary
// (or is 0: a triangle of no height), the small values are RAISED to 2^-10 of the tile's largest (a larger H only keeps more), which
// bounds the amplification of the margin (2^-16 of the magnitudes) by 2^10.
__global__ void __launch_bounds__(kMGroups) k_cylm_hmax(const uint32_t* __restrict__ hdr, float4* __restrict__ rec) {
	__shared__ uint32_t s_max, s_min;
	__shared__ float s_h[kMGroups];
	if (blockIdx.x >= hdr[6]) return;
	if (threadIdx.x == 0) { s_max = 0u; s_min = 0x7f800000u; }
	float4* tile = rec + (size_t)blockIdx.x * kMTileQ;
	const uint32_t grp = threadIdx.x, tb = grp >> 3, j = (grp & 7u) >> 1, hh = grp & 1u;
	const float S = ((const float*)hdr)[7];
	const float k = S > 0.0f ? 256.0f / S : 1.0f;                                       // exact: S is a power of two
	const float4 a = tile[cylm_slot(grp, 4u)], b = tile[cylm_slot(grp, 5u)];           // (Mz H Mz H) of triangles 0,1 and 2,3
	s_h[grp] = fmaxf(fmaxf(a.y, a.w), fmaxf(b.y, b.w)) * k;                             // the group of four; +-inf stay +-inf
	__syncthreads();
	// octets: this quad and its partner (grp ^ 2: the other j of the same q and lane half); both threads hold the same value
	float hm = kMGrp == 8u ? fmaxf(s_h[grp], s_h[grp ^ 2u]) : s_h[grp];
	const bool regular = hm >= 0.0f && hm < __builtin_inff();                           // (positive floats order as their bits)
	if (regular) atomicMax(&s_max, __float_as_uint(hm));
	__syncthreads();
	const float floor_ = fmaxf(__uint_as_float(s_max) * 0x1p-10f, 0x1p-60f);
	if (regular) { hm = fmaxf(hm, floor_); atomicMin(&s_min, __float_as_uint(hm)); }
	__syncthreads();
	// the float4 of (fragment, lane half): component j of a quad; components q = 0, 1 of an octet (written by its first quad: j even)
	if (kMGrp == 4u) ((float*)(tile + cylm_hmq(tb, hh)))[j] = hm;
	else if ((j & 1u) == 0u) ((float*)(tile + cylm_hmq(tb, hh)))[j >> 1] = hm;
	if (threadIdx.x == 0) {
		const float hmin = s_min == 0x7f800000u ? 1.0f : __uint_as_float(s_min);        // no regular octet: any kappa will do
		tile[cylm_kq()] = make_float4((1.0f / hmin) * (1.0f + 0x1p-20f), 0.0f, 0.0f, 0.0f);   // rounded up whatever the division does
	}
}

SP_DEV void cylm_tile_dma(const float4* __restrict__ src, float4* dst, uint32_t tid, uint32_t wbase) {
	typedef __attribute__((address_space(1))) const void* gptr_t;
	typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
	for (int p = 0; p < (int)(kMTileQ / kMThreads); ++p)
		__builtin_amdgcn_global_load_lds((gptr_t)(src + p * kMThreads + tid), (lptr_t)(dst + p * kMThreads + wbase), 16, 0, 0);
}

struct CylmGroup { float4 q0[4]; float4 mh01, mh23; };
SP_DEV CylmGroup cylm_group(const float4* tile, uint32_t grp) {
	CylmGroup G;
#pragma unroll
	for (int u = 0; u < 4; ++u) G.q0[u] = tile[cylm_slot(grp, (uint32_t)u)];
	G.mh01 = tile[cylm_slot(grp, 4u)]; G.mh23 = tile[cylm_slot(grp, 5u)];
	return G;
}

// ---- test-only (sphip_selftest_device, what = 6): the side product of ONE (triangle, ray) pair exactly as stage 1 forms it.
// in: 12 floats per item: the five scaled triangle values (16 b, 16 c, 16 Mc'/S), the five scaled ray values (P_b, P_c, -dir), the ray's
// P_a (a half), 0.  One wave per item; out: 2 floats: the matrix instruction's result, and the same 16 products summed in double.
__global__ void __launch_bounds__(64) k_selftest_cylm(const float* __restrict__ in, uint32_t n, float* __restrict__ out) {
	const uint32_t i = blockIdx.x, lane = threadIdx.x, hh = lane >> 5;
	if (i >= n) return;
	const float* q = in + 12 * (size_t)i;
	_Float16 th[5], tl[5], rh[5], rl[5];
#pragma unroll
	for (int k = 0; k < 5; ++k) { half_split(q[k], th[k], tl[k]); half_split(q[5 + k], rh[k], rl[k]); }
	const _Float16 ah = to_half(q[10]), z = (_Float16)0.0f, one16 = (_Float16)16.0f;
	const half8 a0 = { th[0], tl[0], th[0], th[1], tl[1], th[1], th[2], tl[2] }, a1 = { th[2], th[3], tl[3], th[3], th[4], tl[4], th[4], one16 };
	const half8 b0 = { rh[0], rh[0], rl[0], rh[1], rh[1], rl[1], rh[2], rh[2] }, b1 = { rl[2], rh[3], rh[3], rl[3], rh[4], rh[4], rl[4], ah };
	const half8 zero8 = { z, z, z, z, z, z, z, z };
	const bool mine = (lane & 31u) == 0u;                       // row 0 / column 0 carry the item, everything else is zero
	const half8 af = mine ? (hh ? a1 : a0) : zero8, bf = mine ? (hh ? b1 : b0) : zero8;
	float16v c;
#pragma unroll
	for (int k = 0; k < 16; ++k) c[k] = 0.0f;
	const float16v g = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, c, 0, 0, 0);
	if (lane == 0) {
		double sum = 0.0;
#pragma unroll
		for (int k = 0; k < 8; ++k) sum += (double)(float)a0[k] * (double)(float)b0[k] + (double)(float)a1[k] * (double)(float)b1[k];
		out[2 * (size_t)i] = g[0];
		out[2 * (size_t)i + 1] = (float)sum;
	}
}

// the stream's class table in scalar registers, read ONCE per scan: inside the tile loop a read of cs.hdr is a vector-memory load with
// an s_waitcnt vmcnt(0) behind it -- a round trip to L2 per tile, and a wait for the LDS-DMA in flight as well (vmcnt counts in order)
struct CylmHdr {
	uint32_t n0, n1, n2, f1, f2, tiles;      // triangles per class; first tile of classes 1 and 2 (class 0 starts at tile 0); tiles in all
	uint32_t nbig;                           // triangles of the big class (outside the stream)
	float S;
	// the class being scanned (scalars on purpose: an array indexed by the class number ends up in scratch memory, and a scratch
	// load waits on vmcnt like any other)
	uint32_t cls, cn, cfirst, cend;
	SP_DEV void load(const uint32_t* __restrict__ hdr) {
		n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[0]); n1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[1]);
		n2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[2]);
		f1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[4]); f2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[5]);
		tiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[6]);
		S = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[7]));
		nbig = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[8]);
		cls = 0u; cn = n0; cfirst = 0u; cend = f1;
	}
	// tile gt lies beyond the current class: move on (true), at most twice per tile (an empty class 1)
	SP_DEV bool next_class(uint32_t gt) {
		if (cls >= 2u || gt < cend) return false;
		++cls;
		if (cls == 1u) { cn = n1; cfirst = f1; cend = f2; } else { cn = n2; cfirst = f2; cend = tiles; }
		return true;
	}
	// triangles of the current class in its tile gt, as fragments of 32
	SP_DEV uint32_t fragments(uint32_t gt) const {
		const uint32_t left = cn - (gt - cfirst) * kMTile;
		return ((left < kMTile ? left : kMTile) + 31u) / 32u;
	}
};

// ---- ray side of stage 1.  The lane's own ray: the f32 filter state of cyl_setup (stage 2 recomputes x in f32) and its scaled
// values; per class the B fragments, D^ and Dq^ of the wave's two ray blocks (lane l: column l & 31 of block rb, K half l >> 5)
struct CylmRay {
	CylRay<1> f;               // f32 state, rotated class by class (Pa = the class's dominant axis)
	float Pw[3], Nw[3];        // scaled moment and -dir in the world frame
	float Dh, Dqh;
	bool on_m;
	half8 bfr[2];
	float Dn[2], Dqn[2];

	SP_DEV void setup(float rv, float S, const RaySlots<1>& s) {
		cyl_setup<1>(rv, s, f);
		const float adx = fabsf(s.dir[0].x), ady = fabsf(s.dir[0].y), adz = fabsf(s.dir[0].z);
		const float on = fabsf(s.o[0].x) + fabsf(s.o[0].y) + fabsf(s.o[0].z);
		const float dmax = fmaxf(adx, fmaxf(ady, adz));
		// cyl_setup's verdict (filter on: finite margin) plus the range of the halves; off: zero operands and a margin of +inf
		const bool on0 = (f.Dq[0] < __builtin_inff()) && (f.Dq[0] > 0.0f) && (S > 0.0f) && (on + 2.0f * rv <= 512.0f * S);
		const uint32_t ed = (__float_as_uint(dmax) >> 23) & 255u;
		const float sd0 = on0 ? __uint_as_float((254u - ed) << 23) : 0.0f;                  // 2^-(e-127): dmax * s_d in [1, 2)
		// the scale factors themselves must stay in range: a 1e-30 scene traversed with 1e-18 directions would make them overflow
		// (cyl_setup's guards bound |dir| and |dir|(|pos| + 2Rv) from above, not Rv from below against |dir|)
		const float kP0 = on0 ? sd0 * (16.0f / S) : 0.0f, kC0 = on0 ? sd0 * (256.0f / S) : 0.0f;
		on_m = on0 && (kC0 < 1e30f) && (kP0 > 1e-30f);
		const float s_d = on_m ? sd0 : 0.0f;
		const float kP = on_m ? kP0 : 0.0f, kN = 16.0f * s_d, kC = on_m ? kC0 : 0.0f;
		Dh = on_m ? f.D[0] * s_d : 1.0f;
		// inactive lane: Dq = -inf (rejected); filter off: +inf; else scaled (exact)
		Dqh = on_m ? f.Dq[0] * kC : (s.act[0] ? __builtin_inff() : -__builtin_inff());
		// cyl_setup leaves class 0: (Pa, Pb, Pc) = (P.x, P.y, P.z); build() picks (a, b, c) = (x, y, z), (y, z, x), (z, x, y)
		Pw[0] = f.Pa[0] * kP; Pw[1] = f.Pb[0] * kP; Pw[2] = f.Pc[0] * kP;
		Nw[0] = f.ndx[0] * kN; Nw[1] = f.ndy[0] * kN; Nw[2] = f.ndz[0] * kN;
	}

	// entering class cls (wave-uniform); the f32 state is rotated by the caller
	SP_DEV void build(uint32_t cls, uint32_t lane) {
		const uint32_t hh = lane >> 5;
		const float Pa_ = cls == 0u ? Pw[0] : cls == 1u ? Pw[1] : Pw[2];
		const float Pb_ = cls == 0u ? Pw[1] : cls == 1u ? Pw[2] : Pw[0], Pc_ = cls == 0u ? Pw[2] : cls == 1u ? Pw[0] : Pw[1];
		// scale the ray by t so that its P_a is exactly a half
		const _Float16 ah = to_half(Pa_);
		const float fa = (float)ah;
		const bool big = fabsf(Pa_) >= 0x1p-10f;
		const float t = big ? fa / Pa_ : 1.0f;
		const float extra = big ? 0.0f : fabsf(Pa_ - fa) * 16.0f;                        // the slot's weight on the triangle side is 16
		_Float16 bh, bl, ch, cl, nh[3], nl[3];
		half_split(Pb_ * t, bh, bl); half_split(Pc_ * t, ch, cl);
		half_split(Nw[0] * t, nh[0], nl[0]); half_split(Nw[1] * t, nh[1], nl[1]); half_split(Nw[2] * t, nh[2], nl[2]);
		const half8 k0 = { bh, bh, bl, ch, ch, cl, nh[0], nh[0] };
		const half8 k1 = { nl[0], nh[1], nh[1], nl[1], nh[2], nh[2], nl[2], ah };
		const float Dt = Dh * t;                                                         // D carries 2^-21 of slack: a rounding costs 2^-24
		const float Dqt = on_m ? (Dqh * t + extra) * (1.0f + 0x1p-20f) : Dqh;            // +-inf as they are
		typedef int int4v __attribute__((ext_vector_type(4)));
		const int4v v0 = __builtin_bit_cast(int4v, k0), v1 = __builtin_bit_cast(int4v, k1);
#pragma unroll
		for (int rb = 0; rb < 2; ++rb) {
			const int src = (int)((lane & 31u) + 32u * (uint32_t)rb);
			int4v o;
#pragma unroll
			for (int c = 0; c < 4; ++c) { const int va = __shfl(v0[c], src, 64), vb = __shfl(v1[c], src, 64); o[c] = hh ? vb : va; }
			bfr[rb] = __builtin_bit_cast(half8, o);
			Dn[rb] = __shfl(Dt, src, 64);
			Dqn[rb] = __shfl(Dqt, src, 64);
		}
	}
};

// min(|a|, |b|, |c|, |d|) of four matrix-pipe results.  The results are finite by construction (sums of 16 products of halves), which
// the compiler cannot know: written with fminf it quiets two of the four inputs first (v_max_f32 x, |x|, |x|: 16 more instructions
// per fragment), because v_min_f32 in IEEE mode does not return the other operand for a signalling NaN.  gfx950's IEEE-754-2019
// minimum (v_minimum3_f32: NaN in, NaN out) needs no quieting: two instructions for four values.
#ifndef SP_CYLM_MIN
#define SP_CYLM_MIN 1
#endif
SP_DEV float min4_abs(float a, float b, float c, float d) {
#if SP_CYLM_MIN == 1
	return __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c)), __builtin_fabsf(d));
#elif SP_CYLM_MIN == 2      // median of (|a|, |b|, 0) = min(|a|, |b|): v_med3_f32 is not quieted either
	return __builtin_fminf(__builtin_fminf(__builtin_amdgcn_fmed3f(__builtin_fabsf(a), __builtin_fabsf(b), 0.0f), __builtin_fabsf(c)), __builtin_fabsf(d));
#else
	return __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c)), __builtin_fabsf(d));
#endif
}

// D' of the wave's two ray blocks for tile `cur`: D^ + kappa Dq^, rounded up (D^ > 0; Dq^ > 0, or +inf: filter off -> D' = +inf: every
// regular group survives; or -inf: idle lane -> D' = -inf: none does).  Two instructions per ray block and TILE instead of one per group.
SP_DEV void cylm_tile_bound(const float4* cur, const CylmRay& R, float (&Dt)[2]) {
	const float kappa = cur[cylm_kq()].x;
#pragma unroll
	for (int rb = 0; rb < 2; ++rb) Dt[rb] = __builtin_fmaf(R.Dqn[rb], kappa, R.Dn[rb]) * (1.0f + 0x1p-22f);
}

// min |.| of eight matrix-pipe results: four v_minimum3_f32 (see min4_abs for why not fminf)
SP_DEV float min8_abs(float a, float b, float c, float d, float e, float f, float g, float h) {
	typedef float T;
	const T m1 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c));
	const T m2 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_fabsf(d), __builtin_fabsf(e)), __builtin_fabsf(f));
	const T m3 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(m1, __builtin_fabsf(g)), __builtin_fabsf(h));
	return __builtin_elementwise_minimum(m3, m2);
}

// the VALU part of stage 1 for one fragment and ray block: the survivor bits from the 16 side products g and the Hmax^ of the lane half --
// 4 quad bits (j = 0..3: rows 4 j .. 4 j + 3 of the lane's 16) or 2 octet bits (q = 0, 1: rows 8 q .. 8 q + 7).  x = fma(-Hmax^, D', min_i |g_i|)
// is rounded ONCE, so its sign is the sign of the exact value: v_minimum3 x 2, v_fma, v_alignbit per quad (1 per pair), v_minimum3 x 4, v_fma,
// v_alignbit per octet (0.75 per pair).  Hmax^ > 0 or +-inf (k_cylm_hmax), D' != 0: no 0 x inf.
SP_DEV uint32_t cylm_bits(const float16v& g, const float4 Hm, float Dt, uint32_t word) {
	if constexpr (kMGrp == 4u) {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const float hm = j == 0 ? Hm.x : j == 1 ? Hm.y : j == 2 ? Hm.z : Hm.w;
			const float m = min4_abs(g[4 * j + 0], g[4 * j + 1], g[4 * j + 2], g[4 * j + 3]);
			word = __builtin_amdgcn_alignbit(word, __float_as_uint(__builtin_fmaf(-hm, Dt, m)), 31);
		}
	} else {
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const float hm = q == 0 ? Hm.x : Hm.y;
			const float m = min8_abs(g[8 * q + 0], g[8 * q + 1], g[8 * q + 2], g[8 * q + 3], g[8 * q + 4], g[8 * q + 5], g[8 * q + 6], g[8 * q + 7]);
			word = __builtin_amdgcn_alignbit(word, __float_as_uint(__builtin_fmaf(-hm, Dt, m)), 31);
		}
	}
	return word;
}

// ---- stage 1 of a whole tile (nblk fragments), software-pipelined by hand: the matrix instruction of the NEXT (fragment, ray
// block) is issued before the 20 VALU instructions that turn the previous result into bits, and the LDS reads run a fragment
// ahead -- an in-order wave otherwise sits through the LDS latency, both matrix instructions and their result latency before its
// first VALU instruction of every fragment.  No extra accumulators: a ray block's 16 registers are free again when its bits are out.
SP_DEV void cylm_stage1(const float4* cur, uint32_t tb0, uint32_t nblk, uint32_t lane, const CylmRay& R, const float (&Dt)[2], uint32_t (&word)[2]) {
	if (nblk == 0u) return;
	const half8* frags = (const half8*)(cur + kMRecQ) + tb0 * 64u + lane;
	const float4* hmq = cur + cylm_hmq(tb0, lane >> 5);
	float16v zero;
#pragma unroll
	for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
	// two fragments per trip, their operands in two named sets (A, B): no register copies from "next" to "current", one pair of address
	// increments and one branch per two fragments
#define SP_S1_STEP(afr_cur_Hm, afr_next)                                                     \
		__builtin_amdgcn_sched_barrier(0);                                                   \
		word[0] = cylm_bits(g0, afr_cur_Hm, Dt[0], word[0]);                                 \
		__builtin_amdgcn_sched_barrier(0);                                                   \
		g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr_next, R.bfr[0], zero, 0, 0, 0);      \
		__builtin_amdgcn_sched_barrier(0);                                                   \
		word[1] = cylm_bits(g1, afr_cur_Hm, Dt[1], word[1]);                                 \
		__builtin_amdgcn_sched_barrier(0);                                                   \
		g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr_next, R.bfr[1], zero, 0, 0, 0);
	half8 afrA = frags[0], afrB;
	float4 HmA = hmq[0], HmB;
	float16v g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrA, R.bfr[0], zero, 0, 0, 0);
	float16v g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrA, R.bfr[1], zero, 0, 0, 0);
	uint32_t tb = 1;
	for (; tb + 1u < nblk; tb += 2u) {
		afrB = frags[tb * 64u]; HmB = hmq[tb * 2u];
		SP_S1_STEP(HmA, afrB)
		afrA = frags[(tb + 1u) * 64u]; HmA = hmq[(tb + 1u) * 2u];
		SP_S1_STEP(HmB, afrA)
	}
	if (tb < nblk) {
		afrB = frags[tb * 64u]; HmB = hmq[tb * 2u];
		SP_S1_STEP(HmA, afrB)
		HmA = HmB;
	}
#undef SP_S1_STEP
	word[0] = cylm_bits(g0, HmA, Dt[0], word[0]);
	word[1] = cylm_bits(g1, HmA, Dt[1], word[1]);
}

constexpr uint32_t kMCap = 384u;             // list entries per wave and pass (16 bits each: group << 8 | ray << 2); what does not fit waits for the next pass
constexpr uint32_t kMQ2 = 128u;              // exact-test candidates a wave can hold (32 bits each: ray << 26 | triangle index)
constexpr uint32_t kMIdxBits = 26u;          // hence at most 2^26 - 1 triangles for this scan (the host picks another one beyond)

// value of lane (addr4 / 4) of the wave: ds_bpermute_b32 on a byte address the caller already holds (__shfl would rebuild it: and, or, shift)
SP_DEV float wave_fetch(uint32_t addr4, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)addr4, __float_as_int(v))); }
SP_DEV int wave_fetch(uint32_t addr4, int v) { return __builtin_amdgcn_ds_bpermute((int)addr4, v); }

// inclusive prefix sum over the 64 lanes of a wave: Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8), then the row
// totals (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  Six v_add_u32_dpp, no LDS.  Full waves only.
SP_DEV uint32_t wave_incl_scan(uint32_t x) {
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
	return x;
}

// Closest hit for the ray of every lane.  Block-uniform call (barriers inside).
//
// Stage 2 in two steps, both shared by the whole wave:
//   (a) per tile: the set group bits become a list of (ray, group) entries (list space by a wave prefix sum); 64 entries per round,
//       one per lane: the f32 cylinder test of sp_cyl_scan.h on the group's four records, each with its own H -> candidates;
//   (b) the candidates (ray, triangle index) go to a per-wave stack that OUTLIVES the tile -- the exact test reads the
//       exact record from L2 and the ray from the donor lane's registers, nothing of the tile -- and ray_tri_strict runs only on
//       full batches of 64 (and on what is left at the end of the scan): every lane busy in every exact turn, where one turn
//       per round and its stragglers' turns used half of them (profiles/filter_stats.json).
// Results meet in the 64-bit LDS atomicMin on (d, index) keys, so no order matters (sp_cyl_scan.h, scan_cylw).
SP_DEV void scan_cylm(const KArgs& a, const CylStream cs, float rv, const RaySlots<1>& s, float (&bd)[1], int (&bi)[1]) {
	// Two tile buffers as two separate LDS objects, the tile loop unrolled over them: the LDS-DMA of the next tile is issued BEFORE
	// stage 1 of the current one, and hipcc, which orders every LDS read it cannot tell apart from an outstanding LDS-DMA behind
	// s_waitcnt vmcnt(0), can tell these apart -- the transfer has the whole tile to land in, not just stage 2
	__shared__ float4 sm0[kMTileQ];
	__shared__ float4 sm1[kMTileQ];
	__shared__ unsigned short lst[kMWaves * kMCap];
	__shared__ uint32_t q2[kMWaves * kMQ2];
	__shared__ unsigned long long cell[kMThreads];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wbase = tid & ~63u, hh = lane >> 5;
	unsigned short* const mylst = lst + (tid >> 6) * kMCap;
	uint32_t* const myq2 = q2 + (tid >> 6) * kMQ2;
	CylmHdr hd;
	hd.load(cs.hdr);

	CylmRay R;
	R.setup(rv, hd.S, s);
	CylRay<1>& f = R.f;

	// ---- the big class first: every lane, its own ray, records through the scalar cache (wave-uniform addresses)
	{
		float bd0 = kMaxDist; int bi0 = -1;
		typedef float f4v __attribute__((ext_vector_type(4)));
		typedef __attribute__((address_space(4))) const f4v* cptr_t;             // constant address space: a uniform load is a scalar load
		const cptr_t big = (cptr_t)(uintptr_t)cs.big;
		for (uint32_t b = 0; b < hd.nbig; ++b) {
			const f4v x0 = big[3 * b + 0], x1 = big[3 * b + 1], x2 = big[3 * b + 2];
			const int idx = (int)__float_as_uint(x2.y);
			const float d = ray_tri_strict(s.o[0], s.dir[0], mk3(x0.x, x0.y, x0.z), mk3(x0.w, x1.x, x1.y), mk3(x1.z, x1.w, x2.x));
			// the (d, index) minimum: cpu_renderer.cpp:44's "first strictly smaller d wins" over ascending indices
			const bool take = (d > 0.0f) && (idx != s.src[0]) && ((d < bd0) || (d == bd0 && idx < bi0));
			bd0 = take ? d : bd0;
			bi0 = take ? idx : bi0;
		}
		cell[tid] = ((unsigned long long)__float_as_uint(bd0) << 32) | (unsigned long long)(uint32_t)bi0;     // bi0 = -1: the "no hit" key
	}
	R.build(0u, lane);

	// ---- step (b): exact tests of the top n (<= 64) candidates of the stack; wave-uniform n
	uint32_t q2n = 0;                                     // candidates on the stack (wave-uniform)
	auto exact_batch = [&](uint32_t n) {
#ifdef SP_EXP_NO_EXACT          // timing experiment only (wrong images): the candidates are counted and dropped
		q2n -= n;
		return;
#endif
		const bool ok = lane < n;
		// idle lanes (only in the last batch of a scan): their own ray against the zero record behind the last triangle
		const uint32_t e = ok ? myq2[q2n - n + lane] : ((lane << kMIdxBits) | a.n_tris);
		const uint32_t la = (e >> (kMIdxBits - 2u)) & 0xfcu;                             // 4 x donor lane: the ds_bpermute address
		const int idx = (int)(e & ((1u << kMIdxBits) - 1u));
		// 48 idx by shift and add (v_mul_lo_u32 is a quarter-rate instruction); idx < 2^26: the byte offset fits 32 bits
		const float4* const rec = (const float4*)((const char*)a.scan + ((((uint32_t)idx << 1) + (uint32_t)idx) << 4));
		const float4 x0 = rec[0], x1 = rec[1], x2 = rec[2];
		const float ox = wave_fetch(la, s.o[0].x), oy = wave_fetch(la, s.o[0].y), oz = wave_fetch(la, s.o[0].z);
		const float dx = wave_fetch(la, s.dir[0].x), dy = wave_fetch(la, s.dir[0].y), dz = wave_fetch(la, s.dir[0].z);
		const int src = wave_fetch(la, s.src[0]);
		const float d = ray_tri_strict(mk3(ox, oy, oz), mk3(dx, dy, dz), mk3(x0.x, x0.y, x0.z), mk3(x0.w, x1.x, x1.y), mk3(x1.z, x1.w, x2.x));
		// cpu_renderer.cpp:44: cur_d > 0 && cur_d < d, d starting at MAX_VALUE_DIST; ties -> lowest index: the key's low word
		if (ok && (d > 0.0f) && (d < kMaxDist) && (idx != src))
			atomicMin(&cell[wbase + (la >> 2)], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(uint32_t)idx);
		q2n -= n;
#ifdef SP_FILTER_STATS
		if (lane == 0) { atomicAdd(a.scans + 5, 1ull); atomicAdd(a.scans + 4, (unsigned long long)n); }
#endif
	};

	const uint32_t total_tiles = hd.tiles;
#ifdef SP_EXP_NO_STAGE2
	uint32_t exp_acc = 0;
#endif
#ifdef SP_PHASE_TIMERS          // diagnostic build only (tools/phase_timers.py): wave lifetime by phase, in shader cycles -> a.scans[8..13]
	unsigned long long ph_s1 = 0, ph_list = 0, ph_retest = 0, ph_exact = 0, ph_bar = 0, ph_t0 = 0, ph_t1 = 0;
#define SP_PH_STAMP(v) v = __builtin_amdgcn_s_memtime()
#define SP_PH_ADD(acc, a_, b_) acc += (b_) - (a_)
#else
#define SP_PH_STAMP(v)
#define SP_PH_ADD(acc, a_, b_)
#endif
	__syncthreads();                                  // readers of the previous scan are done with the buffers
	cylm_tile_dma(cs.rec, sm0, tid, wbase);
	__syncthreads();
	auto tile_body = [&](auto parity, const uint32_t gt) {
		const float4* const cur = decltype(parity)::value ? sm1 : sm0;
		float4* const nxt = decltype(parity)::value ? sm0 : sm1;
#ifndef SP_CYLM_LATE_DMA
		// the next tile streams in from here on (its buffer was released by the barrier that ended the previous tile)
		if (gt + 1u < total_tiles) cylm_tile_dma(cs.rec + (size_t)(gt + 1u) * kMTileQ, nxt, tid, wbase);
#endif
		SP_PH_STAMP(ph_t0);
		while (hd.next_class(gt)) {
			// f32 state of stage 2 rotates as in scan_cylw; the halves are rebuilt for the class
			const float t0 = f.Pa[0]; f.Pa[0] = f.Pb[0]; f.Pb[0] = f.Pc[0]; f.Pc[0] = t0;
			R.build(hd.cls, lane);
		}
		const uint32_t nblk = hd.fragments(gt);
		// ---- stage 1: word[w][rb] gets kMBitsFrag bits per fragment kMFragsPerWord w + f (quads j = 0..3: groups 8 tb + 2 j + hh; octets q = 0, 1:
		// groups 8 tb + 4 q + hh and that + 2), first appended = highest
		float Dt[2];
		cylm_tile_bound(cur, R, Dt);
		uint32_t word[kMWords][2];
#pragma unroll
		for (int w = 0; w < (int)kMWords; ++w) {
			word[w][0] = word[w][1] = 0u;
			const uint32_t tb0 = kMFragsPerWord * (uint32_t)w;
			const uint32_t nb = nblk > tb0 ? (nblk - tb0 < kMFragsPerWord ? nblk - tb0 : kMFragsPerWord) : 0u;       // fragments of this word (wave-uniform)
			cylm_stage1(cur, tb0, nb, lane, R, Dt, word[w]);
			const uint32_t done = nb * kMBitsFrag;                    // bits appended; left-align
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) word[w][rb] = done == 0u ? 0u : (word[w][rb] << (32u - done));
#ifdef SP_DBG_ALLBITS
			word[w][0] = word[w][1] = done == 0u ? 0u : (0xffffffffu << (32u - done));
#endif
		}
		SP_PH_STAMP(ph_t1); SP_PH_ADD(ph_s1, ph_t0, ph_t1);
#ifdef SP_CYLM_LATE_DMA
		if (gt + 1u < total_tiles) cylm_tile_dma(cs.rec + (size_t)(gt + 1u) * kMTileQ, nxt, tid, wbase);
#endif
		// ---- stage 2 (a): one list per wave; entry = group << 8 | (ray = donor lane) << 2
#ifdef SP_EXP_NO_STAGE2
		for (int w = 0; w < (int)kMWords; ++w) exp_acc ^= word[w][0] ^ word[w][1];
		for (; false;) {
#else
		for (;;) {
#endif
			uint32_t c = 0;
#pragma unroll
			for (int w = 0; w < (int)kMWords; ++w) c += (uint32_t)__builtin_popcount(word[w][0]) + (uint32_t)__builtin_popcount(word[w][1]);
			const uint32_t incl = wave_incl_scan(c);
			const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
			if (all == 0u) break;
			uint32_t j = incl - c;
			const uint32_t jend = incl < kMCap ? incl : kMCap;            // what does not fit stays in the words for the next pass
			if (all <= kMCap) {
				// everything fits (nearly always): no capacity test per entry, and a plain per-lane loop -- seven vector instructions per entry
#pragma unroll
				for (int w = 0; w < (int)kMWords; ++w)
#pragma unroll
				for (int rb = 0; rb < 2; ++rb) {
					uint32_t m = word[w][rb];
					// bit e of word w: (first) group 64 w + 2 e + hh of a quad, 128 w + 4 e + hh of an octet
					const uint32_t eb = (((lane & 31u) + 32u * (uint32_t)rb) << 2) | ((16u * kMGrp * (uint32_t)w + hh) << 8);
					while (m != 0u) {
						const uint32_t e = (uint32_t)__builtin_clz(m);
						mylst[j++] = (unsigned short)(eb + (e << (kMGrp == 8u ? 10 : 9)));       // (indexed: behind a walking pointer hipcc no longer knows the list from the tile an LDS-DMA is filling)
						m &= ~(0x80000000u >> e);
					}
					word[w][rb] = 0u;
				}
			} else
#pragma unroll
			for (int w = 0; w < (int)kMWords; ++w)
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
				uint32_t m = word[w][rb];
				const uint32_t ray = (lane & 31u) + 32u * (uint32_t)rb;
				while (__any(m != 0u && j < jend)) {
					if (m != 0u && j < jend) {
						const uint32_t e = (uint32_t)__builtin_clz(m);                      // e-th appended bit of word w: fragment 8 w + e / 4, j = e % 4
						mylst[j++] = (unsigned short)((ray << 2) | ((16u * kMGrp * (uint32_t)w + (kMGrp / 2u) * e + hh) << 8));
						m &= ~(0x80000000u >> e);
					}
				}
				word[w][rb] = m;
			}
			__builtin_amdgcn_wave_barrier();                             // every lane's entries are issued before the list is read back (one wave: LDS operations complete in order)
			const uint32_t total = all < kMCap ? all : kMCap;
#ifdef SP_FILTER_STATS
			if (lane == 0) { atomicAdd(a.scans + 1, (unsigned long long)total); atomicAdd(a.scans + 2, (unsigned long long)((total + 63u) / 64u)); }
#endif
			SP_PH_STAMP(ph_t0); SP_PH_ADD(ph_list, ph_t1, ph_t0);
			for (uint32_t base = 0; base < total; base += 64u) {
				const uint32_t ent = base + lane;
				const bool ok = ent < total;
				// entry = group << 8 | 4 ray: the low byte is the ds_bpermute address of the donor lane as it stands
				const uint32_t entry = ok ? (uint32_t)mylst[ent] : (lane << 2);
				const uint32_t la = entry & 0xfcu;
				const uint32_t grp = entry >> 8;
				const float dx = wave_fetch(la, s.dir[0].x), dy = wave_fetch(la, s.dir[0].y), dz = wave_fetch(la, s.dir[0].z);
				const float Pa = wave_fetch(la, f.Pa[0]), Pb = wave_fetch(la, f.Pb[0]), Pc = wave_fetch(la, f.Pc[0]);
				const float D = wave_fetch(la, f.D[0]), Dq = wave_fetch(la, f.Dq[0]);
				// the f32 cylinder test of sp_cyl_scan.h on the entry's kMGrp records (group grp; for an octet grp + 2 as well), each with its own H:
				// which of them survive.  The SIGN of x - Dq, shifted in as in stage 1 (bit kMGrp - 1 - u = triangle u = 4 (second group) + place):
				// one half-rate instruction per triangle instead of compare, mask, select and or.  (survive <=> !(x - Dq >= 0); where the difference
				// is a NaN -- inf - inf: padding records under a ray whose filter is off, or an idle lane -- either answer is right.)
				uint32_t cand = 0;
#pragma unroll
				for (int h = 0; h < (int)(kMGrp / 4u); ++h) {
					const CylmGroup G = cylm_group(cur, grp + 2u * (uint32_t)h);
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const float mz = u == 0 ? G.mh01.x : u == 1 ? G.mh01.z : u == 2 ? G.mh23.x : G.mh23.z;
						const float Hh = u == 0 ? G.mh01.y : u == 1 ? G.mh01.w : u == 2 ? G.mh23.y : G.mh23.w;
						const float x = cyl_x(G.q0[u], mz, Hh, Pa, Pb, Pc, -dx, -dy, -dz, D);
						cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(x - Dq), 31);
					}
				}
				cand = ok ? cand : 0u;
				SP_PH_STAMP(ph_t1); SP_PH_ADD(ph_retest, ph_t0, ph_t1);
				// ---- (b): the candidates go on the wave's stack; a full batch of 64 is tested as soon as the next ones would not fit
				// one pass per candidate RANK, not per place in the octet: a lane's first candidate, then its second, ... (about one candidate
				// per entry: the second pass is short and the later ones hardly ever run).  The triangle index is read from the record (one
				// 4-byte LDS read) when its place is known: bit k = place 3 - (k & 3) of the LAST group minus k >> 2 groups-of-the-entry; the index
				// rows of consecutive groups are consecutive float4, so that of grp + 2 starts 8 words behind that of grp.
				const uint32_t* const gidx = (const uint32_t*)(cur + cylm_slot(grp, 6u));
				const uint32_t lhi = la << (kMIdxBits - 2u);                       // ray << kMIdxBits
				for (;;) {
					const bool mine = cand != 0u;
					const unsigned long long mk = __ballot(mine);
					if (mk == 0ull) break;
					const uint32_t cnt = (uint32_t)__popcll(mk);
					if (q2n + cnt > kMQ2) exact_batch(64u);                  // q2n > kMQ2 - 64 >= 64 here
					const uint32_t k = mine ? (uint32_t)__builtin_ctz(cand) : kMGrp - 1u;
					const uint32_t u = kMGrp - 1u - k;                                  // triangle u of the entry: place u & 3 of its group u >> 2
					const uint32_t idx = gidx[8u * (u >> 2) + (u & 3u)];
					if (mine) myq2[q2n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u))] = lhi | idx;
					q2n += cnt;
					cand &= cand - 1u;
				}
				__builtin_amdgcn_wave_barrier();
				SP_PH_STAMP(ph_t0); SP_PH_ADD(ph_exact, ph_t1, ph_t0);
			}
			SP_PH_STAMP(ph_t1);
		}
		// full batches now rather than later: the exact turns overlap the arrival of the next tile
#ifdef SP_EXP_DRAIN_ALL      // experiment: every wave empties its stack at the end of every tile (same number of exact turns for the waves of a workgroup)
		while (q2n) exact_batch(q2n < 64u ? q2n : 64u);
#else
		while (q2n >= 64u) exact_batch(64u);
#endif
#ifdef SP_FILTER_STATS
		if (lane == 0) atomicAdd(a.scans + 3, 1ull);
#endif
		SP_PH_STAMP(ph_t0); SP_PH_ADD(ph_list, ph_t1, ph_t0);
		__syncthreads();                        // next tile landed (vmcnt(0) in the fence) and this one is free again
		SP_PH_STAMP(ph_t1); SP_PH_ADD(ph_bar, ph_t0, ph_t1);
	};
	for (uint32_t gt = 0; gt < total_tiles; gt += 2u) {
		tile_body(std::integral_constant<int, 0>{}, gt);
		if (gt + 1u < total_tiles) tile_body(std::integral_constant<int, 1>{}, gt + 1u);
	}
	if (q2n) exact_batch(q2n);                  // (q2n < 64 here)
#ifdef SP_PHASE_TIMERS
	if (lane == 0) {
		atomicAdd(a.scans + 8, ph_s1); atomicAdd(a.scans + 9, ph_list); atomicAdd(a.scans + 10, ph_retest); atomicAdd(a.scans + 11, ph_exact);
		atomicAdd(a.scans + 12, ph_bar); atomicAdd(a.scans + 13, 1ull);
	}
#endif
#ifdef SP_EXP_NO_STAGE2
	if (exp_acc == 0x12345678u) cell[tid] = 0ull;
#endif
	__builtin_amdgcn_wave_barrier();
	const unsigned long long k = cell[tid];
	bd[0] = __uint_as_float((uint32_t)(k >> 32));
	bi[0] = (int)(uint32_t)k;
}

// ---- test-only (sphip_selftest_stage1): stage 1 ALONE, exactly as scan_cylm runs it (same ray setup, same fragment function,
// same tiles), for 64 rays per one-wave workgroup (n_rays a multiple of 64).  T = kMTile triangles per tile, W = kMWords words per ray block.
// out_words[(((block * 64 + lane) * tiles + tile) * 2 + rb) * W + w] = the lane's word w.  Quads: bit 31 - (4 f + j) = group 8 (8 w + f) + 2 j + (lane >> 5);
// octets: bit 31 - (2 f + q) = the groups 8 (16 w + f) + 4 q + (lane >> 5) and that + 2 -- of that tile survives for ray 64 block + (lane & 31) + 32 rb.
// out_tri (optional): the same side products g tested PER TRIANGLE with the triangle's own scaled H (x = fma(-H^, D^, |g|), sign(x - Dq^)):
// out_tri[(((block * 64 + lane) * tiles + tile) * 2 + rb) * (T / 64) + tb / 2], bit 31 - (16 (tb & 1) + 4 j + i) = triangle 32 tb + 8 j + 4 (lane >> 5) + i.
// A quad / octet bit may be set where none of its triangle bits is (its bound is weaker), never the other way round.
// Block 0 also writes the stream order: out_order[tile * T + 4 group + u] = triangle index at that place (n_tris = padding).
__global__ void __launch_bounds__(64) k_selftest_stage1(const float* __restrict__ rays, uint32_t n_rays, const CylStream cs, const unsigned int* __restrict__ bounds,
                                                      uint32_t* __restrict__ out_words, uint32_t* __restrict__ out_tri, int* __restrict__ out_order) {
	__shared__ float4 sm[kMTileQ];
	const uint32_t lane = threadIdx.x, hh = lane >> 5;
	const float rv = __uint_as_float(bounds[0]);
	CylmHdr hd;
	hd.load(cs.hdr);
	const uint32_t k = blockIdx.x * 64u + lane, kk = k < n_rays ? k : n_rays - 1u;
	RaySlots<1> s;
	const float* p = rays + (size_t)kk * 6;
	s.o[0] = mk3(p[0], p[1], p[2]); s.dir[0] = mk3(p[3], p[4], p[5]); s.src[0] = -1; s.act[0] = true;
	CylmRay R;
	R.setup(rv, hd.S, s);
	R.build(0u, lane);
	const float kH = hd.S > 0.0f ? 256.0f / hd.S : 1.0f;
	const uint32_t total_tiles = hd.tiles;
	for (uint32_t gt = 0; gt < total_tiles; ++gt) {
		while (hd.next_class(gt)) R.build(hd.cls, lane);
		__syncthreads();
		for (uint32_t q = lane; q < kMTileQ; q += 64u) sm[q] = cs.rec[(size_t)gt * kMTileQ + q];
		__syncthreads();
		const uint32_t nblk = hd.fragments(gt);
		float Dt[2];
		cylm_tile_bound(sm, R, Dt);
		for (uint32_t w = 0; w < kMWords; ++w) {
			uint32_t word[2] = { 0u, 0u };
			const uint32_t tb0 = kMFragsPerWord * w;
			const uint32_t nb = nblk > tb0 ? (nblk - tb0 < kMFragsPerWord ? nblk - tb0 : kMFragsPerWord) : 0u;
			cylm_stage1(sm, tb0, nb, lane, R, Dt, word);
			const uint32_t done = nb * kMBitsFrag;
#pragma unroll
			for (int rb = 0; rb < 2; ++rb)
				if (k < n_rays) out_words[(((size_t)k * total_tiles + gt) * 2u + (uint32_t)rb) * kMWords + w] = done == 0u ? 0u : (word[rb] << (32u - done));
		}
		if (out_tri) {
			float16v zero;
#pragma unroll
			for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
			for (uint32_t tb = 0; tb < kMBlocks; ++tb) {
				const half8 afr = ((const half8*)(sm + kMRecQ))[tb * 64u + lane];
				const float16v g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr, R.bfr[0], zero, 0, 0, 0);
				const float16v g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr, R.bfr[1], zero, 0, 0, 0);
#pragma unroll
				for (int rb = 0; rb < 2; ++rb) {
					const float16v& g = rb == 0 ? g0 : g1;
					uint32_t w = 0;
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						const uint32_t grp = tb * 8u + 2u * (uint32_t)j + hh;
						const float4 a = sm[cylm_slot(grp, 4u)], b = sm[cylm_slot(grp, 5u)];
						const float H4[4] = { a.y * kH, a.w * kH, b.y * kH, b.w * kH };
#pragma unroll
						for (int i = 0; i < 4; ++i) {
							const float x = __builtin_fmaf(-H4[i], R.Dn[rb], __builtin_fabsf(g[4 * j + i]));
							w = __builtin_amdgcn_alignbit(w, __float_as_uint(x - R.Dqn[rb]), 31);
						}
					}
					// two fragments per output word: even fragment in the high half
					if (k < n_rays) {
						uint32_t* o = out_tri + (((size_t)k * total_tiles + gt) * 2u + (uint32_t)rb) * (kMTile / 64u) + (tb >> 1);
						if (tb >= nblk) w = 0u;
						if ((tb & 1u) == 0u) *o = w << 16; else *o |= (w & 0xffffu);
					}
				}
			}
		}
		if (blockIdx.x == 0) {
			for (uint32_t grp = lane; grp < kMGroups; grp += 64u) {
				const float4 gi = sm[cylm_slot(grp, 6u)];
				int* o = out_order + (size_t)gt * kMTile + 4u * grp;
				o[0] = (int)__float_as_uint(gi.x); o[1] = (int)__float_as_uint(gi.y); o[2] = (int)__float_as_uint(gi.z); o[3] = (int)__float_as_uint(gi.w);
			}
		}
	}
}

} } // namespace sp::SP_CYLM_NS
#undef SP_PH_STAMP
#undef SP_PH_ADD
