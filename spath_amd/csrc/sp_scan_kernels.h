// Kernels over the two-stage scans: the closest-hit scan alone, renderer::render_flat, renderer::render.
// SCAN = 0: sp_filter_scan.h (slab filter, per-lane LDS queues)     SCAN = 1: sp_cyl_scan.h (cylinder filter, bit words, stage 2 per lane)
// SCAN = 2: sp_cyl_scan.h scan_cylw (the same stage 1, stage 2 shared by the wave)
// SCAN = 3: sp_cylm_scan.h scan_cylm (stage 1 on the f16 matrix pipe, one ray per lane, stage 2 shared by the wave), 256-thread workgroups
// SCAN = 4: the same scan in 512-thread workgroups with 512-triangle tiles (larger scenes)
#pragma once

#include "sp_filter_scan.h"
#include "sp_cyl_scan.h"
#include "sp_cylm_both.h"

namespace sp {

struct ScanSrc {
	const float4* filt;      // slab records (k_repack_filter)
	CylStream cyl;           // cylinder records (k_cyl_scatter)
	CylStream cylm;          // cylinder records + matrix fragments (k_cylm_scatter), in the tile size of the scan that reads them
};

// threads per workgroup of the kernels below: the matrix-pipe scan shares its (larger) tiles between 8 waves
template <int SCAN> constexpr uint32_t scan_block() { return SCAN == 4 ? cylm512::kMThreads : 256u; }

template <int R, int SCAN>
SP_DEV void two_stage_scan(const KArgs& a, const ScanSrc& src, float rv, const RaySlots<R>& s, float (&bd)[R], int (&bi)[R]) {
	if constexpr (SCAN == 4) { static_assert(R == 1, "scan_cylm: one ray per lane"); cylm512::scan_cylm(a, src.cylm, rv, s, bd, bi); }
	else if constexpr (SCAN == 3) { static_assert(R == 1, "scan_cylm: one ray per lane"); cylm256::scan_cylm(a, src.cylm, rv, s, bd, bi); }
	else if constexpr (SCAN == 2) scan_cylw<R>(a, src.cyl, rv, s, bd, bi);
	else if constexpr (SCAN == 1) scan_cyl<R>(a, src.cyl, rv, s, bd, bi);
	else scan_filter<R>(a, src.filt, rv, s, bd, bi);
}

// ---- the closest-hit scan alone with a two-stage scan; R rays per lane
template <int R, int SCAN>
__global__ void __launch_bounds__(scan_block<SCAN>(), SP_PT_WAVES) k_hit_filter(const KArgs a, const ScanSrc src2, const unsigned int* __restrict__ bounds,
                                                    const int* __restrict__ src_idx, int* __restrict__ out_idx, float* __restrict__ out_d) {
	const float rv = __uint_as_float(bounds[0]);
	RaySlots<R> s;
	uint32_t k[R];
#pragma unroll
	for (int r = 0; r < R; ++r) {
		k[r] = blockIdx.x * (scan_block<SCAN>() * R) + r * scan_block<SCAN>() + threadIdx.x;
		const bool valid = k[r] < a.n_rays;
		const uint32_t kk = valid ? k[r] : a.n_rays - 1;
		const float* p = a.rays + (size_t)kk * 6;
		s.o[r] = mk3(p[0], p[1], p[2]); s.dir[r] = mk3(p[3], p[4], p[5]);
		s.src[r] = src_idx ? src_idx[kk] : -1; s.act[r] = valid;
	}
	float bd[R]; int bi[R];
	two_stage_scan<R, SCAN>(a, src2, rv, s, bd, bi);
	uint32_t nsc = 0;
#pragma unroll
	for (int r = 0; r < R; ++r) if (k[r] < a.n_rays) { out_idx[k[r]] = bi[r]; out_d[k[r]] = bd[r]; nsc++; }
	wave_add_scans(a.scans, nsc);
}

// ---- renderer::render_flat with a two-stage scan; R pixels per lane
template <int R, int SCAN>
__global__ void __launch_bounds__(scan_block<SCAN>(), SP_PT_WAVES) k_flat_filter(const KArgs a, const ScanSrc src2, const unsigned int* __restrict__ bounds) {
	const float rv = __uint_as_float(bounds[0]);
	const uint32_t tid = threadIdx.x;
	RaySlots<R> s;
	uint32_t k[R];
	bool valid[R];
#pragma unroll
	for (int r = 0; r < R; ++r) {
		k[r] = blockIdx.x * (scan_block<SCAN>() * R) + r * scan_block<SCAN>() + tid;
		valid[r] = k[r] < a.n_rays;
		const uint32_t kk = valid[r] ? k[r] : a.n_rays - 1;
		const float* p = a.rays + (size_t)kk * 6;
		s.o[r] = mk3(p[0], p[1], p[2]); s.dir[r] = mk3(p[3], p[4], p[5]);
		s.src[r] = -1; s.act[r] = valid[r];
	}
	float bd[R]; int bi[R];
	two_stage_scan<R, SCAN>(a, src2, rv, s, bd, bi);
	uint32_t nsc = 0;
#pragma unroll
	for (int r = 0; r < R; ++r) {
		uint32_t px = 0;
		if (bi[r] >= 0) {
			const float* m = a.mats + (size_t)bi[r] * 6;
			px = vec3_rgba(mk3(m[0], m[1], m[2]));
		}
		if (valid[r]) { a.out_rgba[k[r]] = px; nsc++; }
	}
	wave_add_scans(a.scans, nsc);
}

// ---- renderer::render with the filter scan; R paths per lane advance in lock-step.
// SPLIT = false: the R slots of a lane are R different pixels (ray k0 + r*256).
// SPLIT = true : the R slots are R CONSECUTIVE SAMPLES of the same pixel (sample smp*R + r); their results are
//                added to the accumulator in sample order, so the sum is the reference's (cpu_renderer.cpp:74-76).
//                A workgroup then covers 256 pixels instead of 256*R: small shards still fill the chip.
// Per-path history (hit index and cos(theta) per depth) and the per-pixel accumulator are parked in a
// global work buffer between scans instead of being held in VGPRs through the scan loop: 52 B per slot,
// touched once per bounce, against ~10^5 VALU instructions per bounce.
//   work layout: hist[depth][k] = {idx, cos bits} (8 B), then acc[c][k] (3 floats), k < n_work
template <int R, bool SPLIT, int SCAN>
__global__ void __launch_bounds__(scan_block<SCAN>(), SP_PT_WAVES) k_pt_filter(const KArgs a, const ScanSrc src2, const unsigned int* __restrict__ bounds,
                                                   int2* __restrict__ hist, float* __restrict__ acc, uint32_t n_work) {
	const float rv = __uint_as_float(bounds[0]);
	const uint32_t tid = threadIdx.x;
	constexpr uint32_t B = scan_block<SCAN>();                // threads per workgroup
	const uint32_t k0 = blockIdx.x * (B * R) + tid;          // work-buffer slot of path r: k0 + r*B
	const uint32_t kstep = SPLIT ? 0u : B;                   // ray index of slot r: kr0 + r*kstep
	const bool chunked = a.n_chunks > 1;                     // blockIdx = chunk * px_blocks + pixel block
	const uint32_t pblk = chunked ? blockIdx.x % a.px_blocks : blockIdx.x;
	const uint32_t chunk = chunked ? blockIdx.x / a.px_blocks : 0u;
	const uint32_t kr0 = SPLIT ? pblk * B + tid : pblk * (B * R) + tid;
	uint32_t pixel[R];
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const uint32_t k = kr0 + r * kstep;
		const uint32_t kk = k < a.n_rays ? k : a.n_rays - 1;
		pixel[r] = (uint32_t)shard_pixel(a, kk);
#pragma unroll
		for (int c = 0; c < 3; ++c) acc[(size_t)c * n_work + k0 + r * B] = 0.0f;
	}
	// primary-hit reuse: cpu_renderer.cpp:74-76 starts every sample from the same vp.rays[idx], so the first
	// scan of all samples of a pixel has one result; the host ran it once per pixel (k_hit_filter) before this launch
	const bool reuse = a.prim_idx != nullptr;
	uint32_t my_scans = 0;
	float pd[R]; int pi[R];
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const uint32_t k = kr0 + r * kstep;
		const uint32_t kk = k < a.n_rays ? k : a.n_rays - 1;
		pd[r] = reuse ? a.prim_d[kk] : 0.0f;
		pi[r] = reuse ? a.prim_idx[kk] : -1;
	}

	const uint32_t n_iter = SPLIT ? (a.n_samples + R - 1) / R : a.n_samples;
	uint32_t it0 = 0, it1 = n_iter;
	if (chunked) {
		const uint32_t per = (n_iter + a.n_chunks - 1) / a.n_chunks;
		it0 = chunk * per < n_iter ? chunk * per : n_iter;
		it1 = it0 + per < n_iter ? it0 + per : n_iter;
	}
	for (uint32_t it = it0; it < it1; ++it) {
		RaySlots<R> s;
		int nh[R];                       // surface hits of this path so far
		uint32_t smp[R];
		bool live[R];                    // this slot carries a sample in this iteration
#pragma unroll
		for (int r = 0; r < R; ++r) {
			const uint32_t k = kr0 + r * kstep;
			smp[r] = SPLIT ? it * R + r : it;
			live[r] = (k < a.n_rays) && (smp[r] < a.n_samples);
			const float* pr = a.rays + (size_t)(k < a.n_rays ? k : a.n_rays - 1) * 6;
			s.o[r] = mk3(pr[0], pr[1], pr[2]); s.dir[r] = mk3(pr[3], pr[4], pr[5]);
			s.src[r] = -1; s.act[r] = live[r];
			nh[r] = 0;
		}
#pragma unroll 1
		for (int depth = 0; depth < 5; ++depth) {
			bool any_alive = false;
#pragma unroll
			for (int r = 0; r < R; ++r) any_alive |= s.act[r];
			if (!__syncthreads_or(any_alive ? 1 : 0)) break;
			float bd[R]; int bi[R];
			if (depth == 0 && reuse) {
#pragma unroll
				for (int r = 0; r < R; ++r) { bd[r] = pd[r]; bi[r] = pi[r]; }
			} else {
				two_stage_scan<R, SCAN>(a, src2, rv, s, bd, bi);
#pragma unroll
				for (int r = 0; r < R; ++r) my_scans += s.act[r] ? 1u : 0u;
			}
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const bool hit = s.act[r] && (bi[r] >= 0);
				if (hit) {
					const float* tn = a.tris + (size_t)bi[r] * 12 + 9;
					f3 n = mk3(tn[0], tn[1], tn[2]);
					if (dot3(n, s.dir[r]) > 0.0f) n = scale3(n, -1.0f);
					double r1, r2;
					philox_uniforms(a.seed, pixel[r], smp[r], (uint32_t)depth, &r1, &r2);
					const f3 nd = rand_unit_vec(n, r1, r2);
					const float ct = dot3(nd, n);
					s.o[r] = add3(s.o[r], scale3(s.dir[r], bd[r]));
					s.dir[r] = nd;
					s.src[r] = bi[r];
					hist[(size_t)depth * n_work + k0 + r * B] = make_int2(bi[r], (int)__float_as_uint(ct));
					nh[r] = depth + 1;
				}
				s.act[r] = hit;
			}
		}
#pragma unroll
		for (int r = 0; r < R; ++r) {
			const uint32_t kw = k0 + r * B;
			f3 rec = mk3(0.0f, 0.0f, 0.0f);
			for (int d = nh[r] - 1; d >= 0; --d) {
				const int2 hc = hist[(size_t)d * n_work + kw];
				const float* m = a.mats + (size_t)hc.x * 6;
				const f3 brdf = scale3(mk3(m[0], m[1], m[2]), kInvPi);
				const f3 e = mk3(m[3], m[4], m[5]);
				rec = add3(e, scale3(scale3(mul3(brdf, rec), __uint_as_float((uint32_t)hc.y)), kInvP));
			}
			// cpu_renderer.cpp:75 accum += sample, in sample order: with SPLIT the slots are consecutive samples of one
			// pixel and are added to slot 0's accumulator one after the other (this loop is unrolled in order)
			if (live[r] && chunked) {
				float* p = a.samp + (size_t)smp[r] * 3 * a.samp_stride + (kr0 + r * kstep);
				p[0] = rec.x; p[a.samp_stride] = rec.y; p[(size_t)2 * a.samp_stride] = rec.z;
			} else if (live[r]) {
				const uint32_t ka = SPLIT ? k0 : kw;
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					float* p = acc + (size_t)c * n_work + ka;
					*p = *p + (c == 0 ? rec.x : c == 1 ? rec.y : rec.z);
				}
			}
		}
	}
#pragma unroll
	for (int r = 0; r < (SPLIT ? 1 : R); ++r) {
		const uint32_t k = kr0 + r * kstep;
		const uint32_t kw = k0 + r * B;
		if (k < a.n_rays && !chunked) {
			const f3 av = scale3(mk3(acc[kw], acc[(size_t)n_work + kw], acc[(size_t)2 * n_work + kw]), a.inv_n);
			a.out_rgba[k] = vec3_rgba(mk3(clamp01(av.x), clamp01(av.y), clamp01(av.z)));
			if (a.out_accum) {
				a.out_accum[(size_t)k * 3 + 0] = av.x;
				a.out_accum[(size_t)k * 3 + 1] = av.y;
				a.out_accum[(size_t)k * 3 + 2] = av.z;
			}
		}
	}
	wave_add_scans(a.scans, my_scans);
}

} // namespace sp
