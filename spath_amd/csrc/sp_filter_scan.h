// Two-stage closest-hit scan ("rpl_filter"): a cheap CONSERVATIVE reject per (ray, triangle),
// the reference's exact Moeller-Trumbore only for the rare survivors.  Bit-identical results.
//
// Stage 1 (every pair, 11 VALU): is the ray outside the slab spanned by the two lines that run parallel
// to one edge of the triangle, one through that edge and one through the opposite vertex?  Written here
// for edge2 (lines through v0 and v1); k_repack_filter picks the longest edge, i.e. the narrowest slab,
// and the same argument holds for the v and u+v slabs.  With w = unit(e2) and
// P = pos x dir (ray moment, per ray):
//     g0 = (pos - v0) . (dir x w) = w.P - dir.(w x v0)        g1 = (pos - v1) . (dir x w) = w.P - dir.(w x v1)
// g0 is the reference's s.h (geom.h:208) up to the factor |e2|, and g0 - g1 its a (geom.h:203).
// geom::ray_intersect accepts only if u = f*(s.h) lies in [0,1] (geom.h:209), i.e. only if g0 and g1
// have opposite signs up to rounding.  So: reject when they have the same sign and the smaller
// magnitude exceeds an error margin Dq.  The record stores the slab's MID-LINE and half-width,
//     Mc = w x (v0 + v1)/2,   h = w x (v1 - v0)/2,      gm = w.P - dir.Mc,   t = dir.h,    g0 = gm + t,  g1 = gm - t,
// which turns "same sign and min(|g0|, |g1|) > Dq" into   |gm| - |t| > Dq   : 9 multiply-adds, one subtract with
// |.| source modifiers, one compare -- no v_med3_f32, which issues at about 60 % of the v_fma_f32 rate on gfx950
// (tools/valu_bench.hip).
//
// Why this never rejects a pair the reference accepts (DESIGN.md section 4 has the full derivation):
// let sh_f, a_f be the floats the strict evaluation produces.  Accepting needs 0 <= fl(fl(1/a_f)*sh_f)
// <= 1, hence sh_f and (a_f - sh_f) have the same sign, or one of them is below 3u*|e1||dir||e2|.
// Every quantity here is a sum of at most 9 products of bounded inputs, so with u = 2^-24
//     |g0*|e2| - sh_f| and |(-g1)*|e2| - (a_f - sh_f)|  <=  36 u |e2| |dir| (|pos| + |v0| + |v1|)
// (strict evaluation 7.6u + stage-1 evaluation 10u: six- and three-term FMA chains and the subtraction + rounding
// of P, Mc, h, w 8u + e1 = fl(v1-v0) 1.8u + the relative slack of the u-comparisons 3u + normalisation 4u, each
// times the magnitude bound).  If g0, g1 share a sign and
// both exceed twice that bound, sh_f and a_f - sh_f provably have opposite signs and exceed the bound:
// the reference rejects.  Dq is set to 2^-16 |dir|_1 (|pos|_1 + 2 Rv) >= 2 * 36u * (...) with the
// 1-norms over-estimating the 2-norms and Rv = max vertex norm of the scene; it is per ray, exact
// |pos| and |dir| of that ray, so there is no assumption on where rays start.  The compare is the
// NaN-safe !(x > Dq): a NaN in stage 1 makes the pair a survivor, and a ray whose magnitudes could make a
// stage-1 product overflow (|dir|_1 (|pos|_1 + 2 Rv) >= 1e37, or NaN) gets Dq = inf: everything survives.
//
// Stage 2: survivors (about 1 % of pairs for small triangles) are queued per lane in LDS and run
// through ray_tri_strict (sp_device_math.h) in index order at the end of each triangle tile, so the
// update rule "first strictly smaller d wins" (cpu_renderer.cpp:44) is preserved.
#pragma once

#include "sp_kernels.h"

#ifndef SP_PT_WAVES
#define SP_PT_WAVES 4      // occupancy target (waves per SIMD) of the path-trace kernel: 4 workgroups of 36 KB LDS per CU
#endif

namespace sp {

#ifndef SP_FLUSH_TILES
#define SP_FLUSH_TILES 1   /* measured: 2 gives +0.8 %, 4 overflows the 24-entry queues (profiles/r01_flush_batching.log) */
#endif
constexpr uint32_t kFlushTiles = SP_FLUSH_TILES;   // tiles between two exact stages
constexpr uint32_t kIdxBits = 10;      // queue entry: slot << kIdxBits | triangle index within the epoch (< kFlushTiles*kTile)
static_assert(kFlushTiles * 256u <= (1u << kIdxBits), "queue index bits");
#ifndef SP_QCAP
#define SP_QCAP 24
#endif
constexpr int kQCap = SP_QCAP;   // queue entries per lane (u16); 12 KB, keeps 4 workgroups per CU

// filter record: 48 B = 3 x float4, produced by k_repack_filter
//   q0 = w.x w.y w.z Mc.x   q1 = Mc.y Mc.z h.x h.y   q2 = h.z 0 0 0
__global__ void __launch_bounds__(256) k_repack_filter(const float* __restrict__ tris, float4* __restrict__ filt, uint32_t n, uint32_t n_padded) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_padded) return;
	if (i >= n) {
		// padding behind the last triangle (the filter loop reads whole groups of 2-4 records): w = 0, Mc = (H,H,H), h = 0
		// gives gm = -H (dx+dy+dz), t = 0: huge, i.e. rejected unless dx+dy+dz is ~0; a padding record that
		// does survive meets a zero exact record (a = 0) in stage 2 and is rejected there
		const float H = 1e30f;
		filt[(size_t)i * 3 + 0] = make_float4(0.0f, 0.0f, 0.0f, H);
		filt[(size_t)i * 3 + 1] = make_float4(H, H, 0.0f, 0.0f);
		filt[(size_t)i * 3 + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		return;
	}
	const float* t = tris + (size_t)i * 12;
	// e1, e2 exactly as the reference rounds them (geom.h:200-201); c = e2 - e1 is the third edge (v1 -> v2)
	const double A[3] = { t[0], t[1], t[2] }, B[3] = { t[3], t[4], t[5] }, C[3] = { t[6], t[7], t[8] };
	const double e1[3] = { (double)(t[3] - t[0]), (double)(t[4] - t[1]), (double)(t[5] - t[2]) };
	const double e2[3] = { (double)(t[6] - t[0]), (double)(t[7] - t[1]), (double)(t[8] - t[2]) };
	const double c[3] = { e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2] };
	const double l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
	const double l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
	const double lc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
	// the narrowest of the three slabs that contain the triangle = the one along its LONGEST edge:
	//   along e2 through v0, apex v1  <->  u in [0,1]      (geom.h:209)
	//   along e1 through v0, apex v2  <->  v in [0,1]      (geom.h:213: v >= 0, and v <= u+v <= 1)
	//   along c  through v1, apex v0  <->  u+v in [0,1]    (geom.h:213: u+v <= 1, and u, v >= 0)
	const double* w3; const double* p0; const double* p1; double len2;
	if (l2 >= l1 && l2 >= lc) { w3 = e2; p0 = A; p1 = B; len2 = l2; }
	else if (l1 >= lc)        { w3 = e1; p0 = A; p1 = C; len2 = l1; }
	else                      { w3 = c;  p0 = B; p1 = A; len2 = lc; }
	const double len = sqrt(len2);
	const float wx = (float)(w3[0] / len), wy = (float)(w3[1] / len), wz = (float)(w3[2] / len);   // NaN for a degenerate triangle: always survives
	const double dwx = wx, dwy = wy, dwz = wz;
	// mid-point and half-difference of the two slab lines' anchor points, in double (exact for float inputs up to
	// 29 binades apart), crossed with w and rounded once
	const double pc[3] = { 0.5 * (p0[0] + p1[0]), 0.5 * (p0[1] + p1[1]), 0.5 * (p0[2] + p1[2]) };
	const double ph[3] = { 0.5 * (p1[0] - p0[0]), 0.5 * (p1[1] - p0[1]), 0.5 * (p1[2] - p0[2]) };
	const float mcx = (float)(dwy * pc[2] - dwz * pc[1]), mcy = (float)(dwz * pc[0] - dwx * pc[2]), mcz = (float)(dwx * pc[1] - dwy * pc[0]);
	const float hx = (float)(dwy * ph[2] - dwz * ph[1]), hy = (float)(dwz * ph[0] - dwx * ph[2]), hz = (float)(dwx * ph[1] - dwy * ph[0]);
	filt[(size_t)i * 3 + 0] = make_float4(wx, wy, wz, mcx);
	filt[(size_t)i * 3 + 1] = make_float4(mcy, mcz, hx, hy);
	filt[(size_t)i * 3 + 2] = make_float4(hz, 0.0f, 0.0f, 0.0f);
}

template <int R>
struct RaySlots {
	f3 o[R], dir[R];
	int src[R];
	bool act[R];
};

// exact closest-hit over the index range [lo, hi) through the scalar path; used on queue overflow
template <int R>
SP_DEV void exact_range(const float4* __restrict__ scan, uint32_t lo, uint32_t hi, const RaySlots<R>& s, float (&bd)[R], int (&bi)[R]) {
	for (uint32_t j = lo; j < hi; ++j) {
		const float4 q0 = scan[3 * j + 0], q1 = scan[3 * j + 1], q2 = scan[3 * j + 2];
		const f3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
#pragma unroll
		for (int r = 0; r < R; ++r) {
			const float d = ray_tri_strict(s.o[r], s.dir[r], v0, e1, e2);
			const bool take = (d > 0.0f) && (d < bd[r]) && ((int)j != s.src[r]);
			bd[r] = take ? d : bd[r];
			bi[r] = take ? (int)j : bi[r];
		}
	}
}

// one 12 KB tile global -> LDS: 3 x 16 B per thread, asynchronous (counts on vmcnt)
SP_DEV void tile_dma(const float4* __restrict__ src, float4* dst, uint32_t tid, uint32_t wbase) {
	typedef __attribute__((address_space(1))) const void* gptr_t;
	typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
	for (int p = 0; p < 3; ++p)
		__builtin_amdgcn_global_load_lds((gptr_t)(src + p * 256 + tid), (lptr_t)(dst + p * 256 + wbase), 16, 0, 0);
}

// stage 1 for one (triangle record, ray): true = survivor
SP_DEV bool slab_survives(const float4 q0, const float4 q1, const float hz, const f3 P, const f3 dir, const float dq) {
	float gm = q0.x * P.x;
	gm = __builtin_fmaf(q0.y, P.y, gm);
	gm = __builtin_fmaf(q0.z, P.z, gm);
	gm = __builtin_fmaf(-dir.x, q0.w, gm);
	gm = __builtin_fmaf(-dir.y, q1.x, gm);
	gm = __builtin_fmaf(-dir.z, q1.y, gm);
	float t = dir.x * q1.z;
	t = __builtin_fmaf(dir.y, q1.w, t);
	t = __builtin_fmaf(dir.z, hz, t);
	return !(fabsf(gm) - fabsf(t) > dq);          // NaN-safe: anything unordered (NaN, inf - inf) survives
}

// Closest hit for the R rays of every lane.  Block-uniform call (barriers inside).
template <int R>
SP_DEV void scan_filter(const KArgs& a, const float4* __restrict__ filt, float rv, const RaySlots<R>& s, float (&bd)[R], int (&bi)[R]) {
	__shared__ float4 sm[2 * kTileQ];
	__shared__ unsigned short qs[kQCap * 256];   // [entry][thread]
	const uint32_t tid = threadIdx.x;
	const uint32_t n_tris = a.n_tris;
	const uint32_t ntiles = (n_tris + kTile - 1) / kTile;

	constexpr uint32_t kU = R >= 4 ? 2u : 4u;   // triangles per filter-loop iteration: one branch per kU x R (= 8) tests
	f3 P[R];
	float Dq[R];
#pragma unroll
	for (int r = 0; r < R; ++r) {
		P[r] = cross3(s.o[r], s.dir[r]);
		const float dn = fabsf(s.dir[r].x) + fabsf(s.dir[r].y) + fabsf(s.dir[r].z);
		const float on = fabsf(s.o[r].x) + fabsf(s.o[r].y) + fabsf(s.o[r].z);
		// every product of stage 1 is bounded by mag, a sum of six by 6 mag: below 1e37 nothing can overflow; beyond
		// (or NaN) the filter is switched off for this ray rather than trusted with infinities
		const float mag = dn * (on + 2.0f * rv);
		const float m = mag < 1e37f ? 0x1p-16f * 1.01f * mag : __builtin_inff();
		// inactive slot: margin -1 -> "|m| > -1" always true -> always rejected.  A non-finite margin
		// (huge or NaN inputs) fails the '>' test for every pair -> everything survives -> exact path.
		Dq[r] = s.act[r] ? m : -1.0f;
		bd[r] = kMaxDist;
		bi[r] = -1;
	}

	// tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write; each wave
	// instruction lands 64 lanes x 16 B contiguously at a wave-uniform LDS base, which is this layout
	const uint32_t wbase = tid & ~63u;
	__syncthreads();                            // readers of the previous scan are done with sm
	tile_dma(filt, sm, tid, wbase);
	__syncthreads();                            // (the barrier's fence waits for the DMA: vmcnt(0))
	uint32_t qn = 0;                              // survivors queued by this lane since the last flush (may exceed kQCap: overflow)
	uint32_t epoch = 0;                           // first tile of the current flush epoch
	for (uint32_t t = 0; t < ntiles; ++t) {
		const float4* cur = sm + (t & 1u) * kTileQ;
		const uint32_t left = n_tris - t * kTile;
		const uint32_t cnt = left < (uint32_t)kTile ? left : (uint32_t)kTile;
		// groups of kU triangles; a ragged tail reads padding records (see k_repack_filter: rejected by stage 1,
		// or failing that by stage 2)
		const uint32_t qend = ((cnt + kU - 1u) / kU) * (3u * kU);
		const uint32_t tile_off = (t - epoch) * kTile;   // index of this tile's first triangle relative to the epoch base
		for (uint32_t q = 0; q < qend; q += 3u * kU) {      // q: float4 offset of the group, wave-uniform (scalar loop)
			float4 a0[kU], a1[kU];
			float a2[kU];
#pragma unroll
			for (int u = 0; u < (int)kU; ++u) { a0[u] = cur[q + 3 * u]; a1[u] = cur[q + 3 * u + 1]; a2[u] = cur[q + 3 * u + 2].x; }
			bool sv[kU][R];
			bool any = false;
#pragma unroll
			for (int u = 0; u < (int)kU; ++u)
#pragma unroll
				for (int r = 0; r < R; ++r) {
					sv[u][r] = slab_survives(a0[u], a1[u], a2[u], P[r], s.dir[r], Dq[r]);
					any |= sv[u][r];
				}
#ifdef SP_ABLATE_NOPUSH
			qn += any ? 1u : 0u;                  // timing experiment only (wrong images): keep the filter alive, skip the queue
			if (false) {
#else
			if (any) {                            // rare: queue the survivors, triangle-major so the order stays ascending
#endif
				const uint32_t j0 = q / 3u;
#pragma unroll
				for (int u = 0; u < (int)kU; ++u)
#pragma unroll
					for (int r = 0; r < R; ++r) if (sv[u][r]) {
						const uint32_t e = qn < (uint32_t)kQCap ? qn : (uint32_t)kQCap - 1u;
						qs[e * 256 + tid] = (unsigned short)((r << kIdxBits) | (tile_off + j0 + u));
						++qn;
					}
			}
		}
		// the next tile streams in while the survivors are processed (issued here, not before the filter loop:
		// the compiler orders every LDS read behind an outstanding LDS-DMA with s_waitcnt vmcnt(0))
		if (t + 1 < ntiles) tile_dma(filt + (size_t)(t + 1) * kTileQ, sm + ((t + 1) & 1u) * kTileQ, tid, wbase);
		// ---- stage 2: exact tests of the survivors queued since the last flush, in queue (= index) order.
		// Flushing every kFlushTiles tiles instead of every tile makes the rounds fuller: a round costs the same
		// whatever the number of lanes that still have an entry, and max-over-lanes of a sum grows slower than the sum.
		if (t + 1 - epoch == kFlushTiles || t + 1 == ntiles) {
#ifdef SP_FILTER_STATS
		{   // experiment build only: survivors, exact rounds, tiles per wave -> a.scans[1..3]
			uint32_t v = qn, mx = qn;
			for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); const uint32_t o2 = __shfl_xor(mx, off, 64); mx = o2 > mx ? o2 : mx; }
			if ((tid & 63u) == 0) { atomicAdd(a.scans + 1, (unsigned long long)v); atomicAdd(a.scans + 2, (unsigned long long)mx); atomicAdd(a.scans + 3, 1ull); if (mx > (uint32_t)kQCap) atomicAdd(a.scans + 4, 1ull); }
		}
#endif
#ifdef SP_ABLATE_NOFLUSH
		if (qn == 0x0fffffffu) bi[0] = 1;         // timing experiment only (wrong images): keep qn alive, skip the exact stage
		qn = 0;
#endif
			const uint32_t base = epoch * kTile;
			if (__builtin_expect(__any(qn > (uint32_t)kQCap), 0)) {
				// some lane overflowed its queue: the whole wave re-scans the epoch exactly (rare: scenes made of
				// triangles so large that most rays cross their slabs)
				const uint32_t hi = (t + 1) * kTile;
				exact_range<R>(a.scan, base, hi < n_tris ? hi : n_tris, s, bd, bi);
			} else {
				for (uint32_t e = 0; __any(e < qn); ++e) {
					if (e < qn) {
						const uint32_t ent = qs[e * 256 + tid];
						const int slot = (int)(ent >> kIdxBits);
						const uint32_t idx = base + (ent & ((1u << kIdxBits) - 1u));
						// pick the slot's ray with explicit per-component selects (a struct copy under `if (slot == r)` made the
						// compiler index the slot array dynamically and spill it to scratch for R = 4)
						float ox = s.o[0].x, oy = s.o[0].y, oz = s.o[0].z, dx = s.dir[0].x, dy = s.dir[0].y, dz = s.dir[0].z;
						int src = s.src[0];
#pragma unroll
						for (int r = 1; r < R; ++r) {
							const bool pick = (slot == r);
							ox = pick ? s.o[r].x : ox; oy = pick ? s.o[r].y : oy; oz = pick ? s.o[r].z : oz;
							dx = pick ? s.dir[r].x : dx; dy = pick ? s.dir[r].y : dy; dz = pick ? s.dir[r].z : dz;
							src = pick ? s.src[r] : src;
						}
						const f3 o = mk3(ox, oy, oz), dir = mk3(dx, dy, dz);
						const float4 x0 = a.scan[3 * idx + 0], x1 = a.scan[3 * idx + 1], x2 = a.scan[3 * idx + 2];
						const float d = ray_tri_strict(o, dir, mk3(x0.x, x0.y, x0.z), mk3(x0.w, x1.x, x1.y), mk3(x1.z, x1.w, x2.x));
#pragma unroll
						for (int r = 0; r < R; ++r) {
							const bool take = (slot == r) && (d > 0.0f) && (d < bd[r]) && ((int)idx != src);
							bd[r] = take ? d : bd[r];
							bi[r] = take ? (int)idx : bi[r];
						}
					}
				}
			}
			qn = 0;
			epoch = t + 1;
		}
		__syncthreads();                        // next tile landed (vmcnt(0) in the fence) and this one is free again
	}
}

} // namespace sp
