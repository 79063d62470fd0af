// Kernels of the spath hot path for gfx950 (CDNA4): triangle repack, flat pass, path tracer.
//
// Work decomposition: one lane owns one pixel for the whole integrator -- or, in a
// sample-chunked launch of the filter kernels, for one contiguous range of its samples, with k_resolve adding
// the per-sample results up -- so samples are accumulated in the reference's order (cpu_renderer.cpp:74-76); a wavefront walks 64
// paths in lock-step through the (wave-uniform) depth loop, and the closest-hit scan streams the
// repacked triangle array once per (wave, bounce).
#pragma once

#include "sp_device_math.h"

namespace sp {

// scan record: 48 B = 3 x float4, produced by k_repack
//   q0 = v0.x v0.y v0.z e1.x   q1 = e1.y e1.z e2.x e2.y   q2 = e2.z 0 0 0
struct KArgs {
	const float*  rays;        // n_rays * 6
	const float4* scan;        // n_tris * 3
	const float*  tris;        // n_tris * 12 (normals live at +9)
	const float*  mats;        // n_tris * 6
	uint32_t*     out_rgba;    // n_rays
	float*        out_accum;   // n_rays * 3 or nullptr
	unsigned long long* scans; // device counter
	uint32_t n_rays, n_tris, n_samples, flags;
	uint64_t seed;
	uint64_t pixel_base, tile_px, tile_stride_px;
	float inv_n;               // float(1.0/n_samples), cpu_renderer.cpp:77
	// sample chunks (filter kernels): blockIdx = chunk * px_blocks + pixel block; every sample's radiance is written to
	// samp[(sample * 3 + c) * samp_stride + ray] and k_resolve adds them up in sample order.  n_chunks <= 1: off
	uint32_t n_chunks, px_blocks, samp_stride;
	float* samp;
	// primary-hit reuse of the two-stage kernels (flags & 0x100): closest hit of every ray of the launch, from a pre-pass
	// (k_hit_filter, one scan per PIXEL); k_pt_filter starts every sample of the pixel from it.  nullptr: off
	const int*   prim_idx;     // n_rays
	const float* prim_d;       // n_rays
};

// second pass of a sample-chunked launch: cpu_renderer.cpp:72-78 for one pixel -- zero, += sample in sample order,
// * float(1.0/n_samples), clamp, quantise
__global__ void __launch_bounds__(256) k_resolve(const KArgs a) {
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	if (k >= a.n_rays) return;
	float ax = 0.0f, ay = 0.0f, az = 0.0f;
	for (uint32_t s = 0; s < a.n_samples; ++s) {
		const float* p = a.samp + (size_t)s * 3 * a.samp_stride + k;
		ax = ax + p[0];
		ay = ay + p[a.samp_stride];
		az = az + p[(size_t)2 * a.samp_stride];
	}
	const f3 av = scale3(mk3(ax, ay, az), a.inv_n);
	a.out_rgba[k] = vec3_rgba(mk3(clamp01(av.x), clamp01(av.y), clamp01(av.z)));
	if (a.out_accum) {
		a.out_accum[(size_t)k * 3 + 0] = av.x;
		a.out_accum[(size_t)k * 3 + 1] = av.y;
		a.out_accum[(size_t)k * 3 + 2] = av.z;
	}
}

// ---- repack: AoS geom::triangle -> scan records.  e1/e2 are the single float subtractions of
// geom.h:200-201, hoisted out of the per-ray test (same bits).  Also the scene bound Rv (bounds[0], zeroed by the host) that the
// margins of the two-stage scans are built from.
__global__ void __launch_bounds__(256) k_repack(const float* __restrict__ tris, float4* __restrict__ scan, unsigned int* __restrict__ bounds, uint32_t n, uint32_t n_padded) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_padded) return;
	if (i >= n) {   // padding record: e1 = e2 = 0 -> a = 0 -> rejected at geom.h:204, can never be hit
		const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		scan[(size_t)i * 3 + 0] = z; scan[(size_t)i * 3 + 1] = z; scan[(size_t)i * 3 + 2] = z;
		return;
	}
	const float* t = tris + (size_t)i * 12;
	const float v0x = t[0], v0y = t[1], v0z = t[2];
	const float e1x = t[3] - v0x, e1y = t[4] - v0y, e1z = t[5] - v0z;
	const float e2x = t[6] - v0x, e2y = t[7] - v0y, e2z = t[8] - v0z;
	scan[(size_t)i * 3 + 0] = make_float4(v0x, v0y, v0z, e1x);
	scan[(size_t)i * 3 + 1] = make_float4(e1y, e1z, e2x, e2y);
	scan[(size_t)i * 3 + 2] = make_float4(e2z, 0.0f, 0.0f, 0.0f);
	// scene bound Rv >= every vertex norm, as 1-norms (>= 2-norm); non-negative floats order like their bit patterns;
	// a NaN or inf coordinate yields a bit pattern >= inf, which turns the filter off in the kernels
	float r = 0.0f;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const float s = fabsf(t[3 * k]) + fabsf(t[3 * k + 1]) + fabsf(t[3 * k + 2]);
		r = (s > r || s != s) ? s : r;
	}
	atomicMax(bounds, __float_as_uint(r) & 0x7fffffffu);
}

// ---- view::camera::get_viewport (view.h:94-132) on the device: one thread per pixel.
// The eight step constants are computed on the host in the reference's mixed double/float way (view.h:101-108).
struct ViewArgs {
	float x_max, x_step, h_x_step, y_max, y_step, h_y_step;
	float focal, cos_y, sin_y, cos_x, sin_x;
	float px, py, pz;
	uint32_t res_x, res_y;
	// which pixels: ray k of the output is global pixel pixel_base + (k / tile_px) * tile_stride_px + k % tile_px (sphip_shard);
	// the whole image is {0, res_x*res_y, 0} with n_local = res_x*res_y
	uint64_t pixel_base, tile_px, tile_stride_px;
	uint32_t n_local;
};

SP_DEV f3 cam_rel_move(const ViewArgs& v, f3 in) {                               // view.h:83-85 = rY(rX(in))
	const f3 a = mk3(in.x, in.y * v.cos_x + in.z * -v.sin_x, in.y * v.sin_x + in.z * v.cos_x);   // :62-68
	return mk3(a.x * v.cos_y + a.z * v.sin_y, a.y, a.x * -v.sin_y + a.z * v.cos_y);              // :54-60
}

__global__ void __launch_bounds__(256) k_viewport(const ViewArgs v, float* __restrict__ rays) {
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	if (k >= v.n_local) return;
	const uint64_t tile = (uint64_t)k / v.tile_px;
	const uint64_t idx = v.pixel_base + tile * v.tile_stride_px + ((uint64_t)k - tile * v.tile_px);
	const int i = (int)(idx % v.res_x), j = (int)(idx / v.res_x);                 // :112 index = i + j*res_x
	const f3 cur = mk3(v.x_max - v.x_step * (float)i - v.h_x_step, v.y_max - v.y_step * (float)j - v.h_y_step, 0.0f);   // :111
	const f3 t = add3(cur, mk3(0.0f, 0.0f, v.focal));                             // :114
	const float l = __builtin_sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);           // geom.h:130-136 (IEEE sqrt)
	const f3 dir = cam_rel_move(v, mk3(t.x / l, t.y / l, t.z / l));               // :138-141 then view.h:127
	const f3 pos = add3(cam_rel_move(v, cur), mk3(v.px, v.py, v.pz));             // view.h:126,131
	float* o = rays + (size_t)k * 6;
	o[0] = pos.x; o[1] = pos.y; o[2] = pos.z; o[3] = dir.x; o[4] = dir.y; o[5] = dir.z;
}

// ---- reassembly of a frame rendered as interleaved row tiles on G devices (one padded buffer of `pad` pixels per device,
// gathered to one device): out[p] = the p-th pixel of the image.  C = dwords per pixel (1: RGBA8, 3: float accumulators)
template <int C>
__global__ void __launch_bounds__(256) k_assemble(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ out, uint32_t npix,
                                                 uint32_t tile_px, uint32_t n_dev, uint32_t pad) {
	const uint32_t p = blockIdx.x * 256u + threadIdx.x;
	if (p >= npix) return;
	const uint32_t tile = p / tile_px, r = tile % n_dev, k = (tile / n_dev) * tile_px + (p - tile * tile_px);
#pragma unroll
	for (int c = 0; c < C; ++c) out[(size_t)p * C + c] = gathered[((size_t)r * pad + k) * C + c];
}

// ---- test-only: the device functions of the path on caller-supplied inputs (include/spath_hip.h: sphip_selftest_device)
__global__ void __launch_bounds__(256) k_selftest(int what, const void* __restrict__ in, uint32_t n, void* __restrict__ out) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	if (what == 0) {
		float sn, cs;
		sincos_glibc(((const float*)in)[i], &sn, &cs);
		((float*)out)[2 * i] = sn; ((float*)out)[2 * i + 1] = cs;
	} else if (what == 1) {
		((float*)out)[i] = recip_ieee(((const float*)in)[i]);
	} else if (what == 2) {
		const uint32_t* q = (const uint32_t*)in + 5 * (size_t)i;
		double r1, r2;
		philox_uniforms((uint64_t)q[0] | ((uint64_t)q[1] << 32), q[2], q[3], q[4], &r1, &r2);
		((double*)out)[2 * i] = r1; ((double*)out)[2 * i + 1] = r2;
	} else if (what == 3) {
		const double* q = (const double*)in + 5 * (size_t)i;
		const f3 v = rand_unit_vec(mk3((float)q[0], (float)q[1], (float)q[2]), q[3], q[4]);
		float* o = (float*)out + 3 * (size_t)i;
		o[0] = v.x; o[1] = v.y; o[2] = v.z;
	} else if (what == 4) {
		const float* q = (const float*)in + 15 * (size_t)i;
		const f3 v0 = mk3(q[6], q[7], q[8]);
		// e1, e2 as k_repack forms them: one float subtraction each (geom.h:200-201)
		((float*)out)[i] = ray_tri_strict(mk3(q[0], q[1], q[2]), mk3(q[3], q[4], q[5]), v0, sub3(mk3(q[9], q[10], q[11]), v0), sub3(mk3(q[12], q[13], q[14]), v0));
	} else if (what == 5) {
		const float* q = (const float*)in + 3 * (size_t)i;
		((uint32_t*)out)[i] = vec3_rgba(mk3(clamp01(q[0]), clamp01(q[1]), clamp01(q[2])));
	}
}

SP_DEV uint64_t shard_pixel(const KArgs& a, uint32_t k) {
	const uint64_t t = (uint64_t)k / a.tile_px;
	return a.pixel_base + t * a.tile_stride_px + ((uint64_t)k - t * a.tile_px);
}

// ---- closest-hit scan, variant "rpl_sload": ray per lane, triangle index wave-uniform so the
// records arrive through the scalar data path (s_load_dwordx4) and every VALU op reads them as
// SGPR operands.  Semantics of cpu_renderer.cpp:36-49: ascending index, strict '<', skip idx_source.
SP_DEV void scan_rpl_sload(const float4* __restrict__ scan, uint32_t n_tris, f3 o, f3 dir, int src,
                           float& best_d, int& best_i) {
	float bd = kMaxDist;
	int bi = -1;
	for (uint32_t j = 0; j < n_tris; ++j) {
		const float4 q0 = scan[3 * j + 0], q1 = scan[3 * j + 1], q2 = scan[3 * j + 2];
		const f3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
		const float d = ray_tri_strict(o, dir, v0, e1, e2);
		const bool take = (d > 0.0f) && (d < bd) && ((int)j != src);
		bd = take ? d : bd;
		bi = take ? (int)j : bi;
	}
	best_d = bd;
	best_i = bi;
}

// ---- closest-hit scan, variant "rpl_lds": ray per lane; the workgroup streams the scan records
// HBM/L2 -> registers -> LDS in coalesced 16-byte pieces (double-buffered tiles of kTile triangles),
// and every lane reads the current triangle from LDS at a wave-uniform address (hardware broadcast),
// so all VALU operands are VGPRs.  One triangle fetched from L2/HBM is shared by the 256 rays of the
// workgroup.  Must be called by every thread of the block.
constexpr int kTile = 256;                      // triangles per LDS tile (12 KB), two tiles in flight
constexpr int kTileQ = kTile * 3;               // float4 per tile
static_assert(kTileQ % 256 == 0, "tile must split evenly over the 256 threads");

SP_DEV void scan_rpl_lds(const float4* __restrict__ scan, uint32_t n_tris, f3 o, f3 dir, int src,
                         float& best_d, int& best_i) {
	__shared__ float4 sm[2 * kTileQ];
	static_assert(kTileQ == 3 * 256, "three float4 per thread per tile");
	const uint32_t tid = threadIdx.x;
	const uint32_t ntiles = (n_tris + kTile - 1) / kTile;   // the scan buffer is zero-padded to whole tiles
	float4 p0 = scan[tid], p1 = scan[256 + tid], p2 = scan[512 + tid];
	__syncthreads();                            // readers of the previous scan are done with sm
	sm[tid] = p0; sm[256 + tid] = p1; sm[512 + tid] = p2;
	__syncthreads();
	float bd = kMaxDist;
	int bi = -1;
	for (uint32_t t = 0; t < ntiles; ++t) {
		const float4* cur = sm + (t & 1u) * kTileQ;
		const bool more = (t + 1 < ntiles);
		const float4* nsrc = scan + (size_t)(more ? t + 1 : t) * kTileQ;   // last tile: harmless re-read
		p0 = nsrc[tid]; p1 = nsrc[256 + tid]; p2 = nsrc[512 + tid];
		const uint32_t left = n_tris - t * kTile;
		const uint32_t cnt = ((left < (uint32_t)kTile ? left : (uint32_t)kTile) + 3u) & ~3u;   // zero records never hit
		const int base = (int)(t * kTile);
#pragma unroll 4
		for (uint32_t j = 0; j < cnt; ++j) {
			const float4 q0 = cur[3 * j + 0], q1 = cur[3 * j + 1], q2 = cur[3 * j + 2];
			const f3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
			const float d = ray_tri_strict(o, dir, v0, e1, e2);
			const int idx = base + (int)j;
			const bool take = (d > 0.0f) && (d < bd) && (idx != src);
			bd = take ? d : bd;
			bi = take ? idx : bi;
		}
		float4* nxt = sm + ((t + 1) & 1u) * kTileQ;   // not read before the barrier below + the next one
		nxt[tid] = p0; nxt[256 + tid] = p1; nxt[512 + tid] = p2;
		__syncthreads();
	}
	best_d = bd;
	best_i = bi;
}

// VARIANT 1 = rpl_sload, 2 = rpl_lds.  Every variant must be called block-uniformly.
template <int VARIANT>
SP_DEV void closest_hit(const KArgs& a, f3 o, f3 dir, int src, float& best_d, int& best_i) {
	if (VARIANT == 2) scan_rpl_lds(a.scan, a.n_tris, o, dir, src, best_d, best_i);
	else scan_rpl_sload(a.scan, a.n_tris, o, dir, src, best_d, best_i);
}

SP_DEV void wave_add_scans(unsigned long long* ctr, uint32_t mine) {
	// one atomic per wavefront
	uint32_t v = mine;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	if ((threadIdx.x & 63u) == 0 && v) atomicAdd(ctr, (unsigned long long)v);
}

// ---- renderer::render_flat (cpu_renderer.cpp:81-101): nearest triangle's reflectance, no skip
template <int VARIANT>
__global__ void __launch_bounds__(256) k_flat(const KArgs a) {
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	const bool valid = k < a.n_rays;
	const uint32_t kk = valid ? k : a.n_rays - 1;
	const float* r = a.rays + (size_t)kk * 6;
	const f3 o = mk3(r[0], r[1], r[2]), dir = mk3(r[3], r[4], r[5]);
	float bd; int bi;
	closest_hit<VARIANT>(a, o, dir, -1, bd, bi);
	uint32_t px = 0;                                             // :89 RGBA{0,0,0,0}
	if (bi >= 0) {
		const float* m = a.mats + (size_t)bi * 6;
		px = vec3_rgba(mk3(m[0], m[1], m[2]));                   // :96
	}
	if (valid) a.out_rgba[k] = px;
	wave_add_scans(a.scans, valid ? 1u : 0u);
}

// ---- the closest-hit scan alone (cpu_renderer.cpp:36-49), one ray per lane
template <int VARIANT>
__global__ void __launch_bounds__(256) k_hit(const KArgs a, const int* __restrict__ src_idx, int* __restrict__ out_idx, float* __restrict__ out_d) {
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	const bool valid = k < a.n_rays;
	const uint32_t kk = valid ? k : a.n_rays - 1;
	const float* r = a.rays + (size_t)kk * 6;
	float bd; int bi;
	closest_hit<VARIANT>(a, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), src_idx ? src_idx[kk] : -1, bd, bi);
	if (valid) { out_idx[k] = bi; out_d[k] = bd; }
	wave_add_scans(a.scans, valid ? 1u : 0u);
}

// ---- renderer::render (cpu_renderer.cpp:29-79): n_samples x (<=5 surface hits).
// The recursion of render_step is run forward (store idx and cos(theta) per depth) and unwound
// backward in the reference's own evaluation order  E + (((BRDF*rec)*cos)*(1/p))  (:67) -- the
// shape the reference itself uses in its GLSL backend (render.comp:160-215).
template <int VARIANT>
__global__ void __launch_bounds__(256) k_pt(const KArgs a) {
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	const bool valid = k < a.n_rays;
	const uint32_t kk = valid ? k : a.n_rays - 1;
	const float* r = a.rays + (size_t)kk * 6;
	const f3 po = mk3(r[0], r[1], r[2]), pdir = mk3(r[3], r[4], r[5]);
	const uint32_t pixel = (uint32_t)shard_pixel(a, kk);
	const bool reuse = (a.flags & 0x100u) != 0;

	uint32_t my_scans = 0;
	// optional primary-hit reuse: the primary ray is the same for every sample (:74-76)
	float pd = 0.0f; int pi = -1;
	if (reuse) { closest_hit<VARIANT>(a, po, pdir, -1, pd, pi); my_scans += valid ? 1u : 0u; }

	f3 accum = mk3(0.0f, 0.0f, 0.0f);
	for (uint32_t s = 0; s < a.n_samples; ++s) {
		f3 o = po, dir = pdir;
		int src = -1;
		int idx0 = -1, idx1 = -1, idx2 = -1, idx3 = -1, idx4 = -1;
		float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f, c4 = 0.0f;
		bool alive = valid;
#pragma unroll 1
		for (int depth = 0; depth < 5; ++depth) {                // :33 depth >= 5 -> black
			if (!__syncthreads_or(alive ? 1 : 0)) break;   // block-uniform: the LDS scan has barriers
			float bd; int bi;
			if (depth == 0 && reuse) { bd = pd; bi = pi; }
			else { closest_hit<VARIANT>(a, o, dir, src, bd, bi); my_scans += alive ? 1u : 0u; }
			const bool hit = alive && (bi >= 0);                  // :51 miss -> black
			if (hit) {
				const float* tn = a.tris + (size_t)bi * 12 + 9;
				f3 n = mk3(tn[0], tn[1], tn[2]);                  // :55
				if (dot3(n, dir) > 0.0f) n = scale3(n, -1.0f);    // :56-57
				double r1, r2;
				philox_uniforms(a.seed, pixel, s, (uint32_t)depth, &r1, &r2);
				const f3 nd = rand_unit_vec(n, r1, r2);           // :58
				const float ct = dot3(nd, n);                     // :62
				o = add3(o, scale3(dir, bd));                     // geom.h:218 point = pos + dir*d
				dir = nd;
				src = bi;
				if (depth == 0) { idx0 = bi; c0 = ct; }
				else if (depth == 1) { idx1 = bi; c1 = ct; }
				else if (depth == 2) { idx2 = bi; c2 = ct; }
				else if (depth == 3) { idx3 = bi; c3 = ct; }
				else { idx4 = bi; c4 = ct; }
			}
			alive = hit;
		}
		// unwind: rec(depth) = E + (((BRDF * rec(depth+1)) * cos) * (1/p)), rec beyond the last hit = 0
		f3 rec = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
		for (int depth = 4; depth >= 0; --depth) {
			const int id = depth == 0 ? idx0 : depth == 1 ? idx1 : depth == 2 ? idx2 : depth == 3 ? idx3 : idx4;
			const float ct = depth == 0 ? c0 : depth == 1 ? c1 : depth == 2 ? c2 : depth == 3 ? c3 : c4;
			if (id >= 0) {
				const float* m = a.mats + (size_t)id * 6;
				const f3 brdf = scale3(mk3(m[0], m[1], m[2]), kInvPi);                     // :63
				const f3 e = mk3(m[3], m[4], m[5]);
				rec = add3(e, scale3(scale3(mul3(brdf, rec), ct), kInvP));                 // :67
			}
		}
		accum = add3(accum, rec);                                // :75
	}
	accum = scale3(accum, a.inv_n);                              // :77
	if (valid) {
		a.out_rgba[k] = vec3_rgba(mk3(clamp01(accum.x), clamp01(accum.y), clamp01(accum.z)));  // :78
		if (a.out_accum) {
			a.out_accum[(size_t)k * 3 + 0] = accum.x;
			a.out_accum[(size_t)k * 3 + 1] = accum.y;
			a.out_accum[(size_t)k * 3 + 2] = accum.z;
		}
	}
	wave_add_scans(a.scans, my_scans);
}

} // namespace sp
