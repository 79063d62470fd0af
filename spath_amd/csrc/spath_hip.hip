// libspath_hip.so -- C ABI (include/spath_hip.h) over the gfx950 kernels in sp_kernels.h.
//
// Host side of the drop-in boundary: owns the device buffers (grow-only, like the reference's
// OpenCL peer caches its cl::Buffers, src/cl_renderer.cpp:107-112), uploads what
// renderer::render / render_flat are handed (src/renderer.h:31-32), launches, reads back.
// No exception crosses this boundary; every failure becomes a status + sphip_last_error().
#include "spath_hip.h"
#include "sp_kernels.h"
#include "sp_filter_scan.h"
#include "sp_cyl_scan.h"
#include "sp_scan_kernels.h"
#include "sp_bvh.h"
#include "sp_bvh_build.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <dlfcn.h>
#include <thread>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
};

} // namespace

struct sphip_ctx {
	int device = 0;
	hipStream_t own_stream = nullptr;       // host-pointer path
	hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_u0 = nullptr, ev_u1 = nullptr, ev_d0 = nullptr, ev_d1 = nullptr;
	DevBuf tris, mats, scan, filt, bounds, samp, rays, rgba, accum, counter, work, bvh_nodes, bvh_rec, bvh_idx, sort_kv, sort_hist, bvh_meta, cyl_rec, cyl_cnt, cyl_hdr, prim, cylm_rec, cylm_hdr, cylm_big;
	// record streams beyond the exact one are derived from `tris` the first time a kernel variant that reads them runs on the scene
	bool bvh_valid = false, filt_valid = false, cyl_valid = false, cylm_valid = false;
	bool cylm_wide = false;                 // the stream in cylm_rec is laid out for the 512-thread shape of the default scan (sp_cylm_both.h)
	uint32_t bvh_leaves = 0;
	size_t n_tris = 0;
	bool have_scene = false;
	bool have_render = false, timed_upload = false, timed_download = false;
	sphip_stats stats{};
	std::string err;
	std::string desc;
	hipStream_t last_stream = nullptr;
	// ---- multi-device context (sphip_create_multi): one child context per listed device; the exchange buffers live on the
	// first child's device.  A single-device context has no kids.
	std::vector<sphip_ctx*> kids;
	int gather_kind = SPHIP_GATHER_NONE;
	void* rccl_lib = nullptr;
	std::vector<void*> comms;                     // ncclComm_t per child
	DevBuf gath, gath_acc, img, img_acc;           // [n_dev][pad] tiles as gathered, and the image in pixel order
	std::vector<hipEvent_t> ev_tile;               // child r's tiles have arrived on the first device
	hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;
};

namespace {

int fail(sphip_ctx* c, int code, const char* fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (c) c->err = buf; else g_create_error = buf;
	return code;
}

#define HIP_TRY(c, expr)                                                                         \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess)                                                                    \
			return fail((c), SPHIP_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

int ensure(sphip_ctx* c, DevBuf& b, size_t bytes) {
	if (bytes <= b.cap && b.p) return SPHIP_OK;
	if (b.p) { HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
	const size_t want = bytes < 256 ? 256 : bytes;
	HIP_TRY(c, hipMalloc(&b.p, want));
	b.cap = want;
	return SPHIP_OK;
}

// kernel variants selectable through the low byte of `flags` (sphip_kernel_name); all brute force except 8
constexpr int kVariantAccel = 8;          // the opt-in acceleration structure (SPHIP_FLAG_ACCEL)
constexpr int kVariantLast = 16;
constexpr uint64_t kChunkTargetBlocks = 262144;         // 256 x the 1024 resident workgroups (measured: profiles/r01_sample_chunks.log)
constexpr uint64_t kChunkMaxBytes = 16ull << 30;         // cap of the per-sample scratch buffer
const char* const kVariantNames[kVariantLast + 1] = { "auto", "rpl_sload", "rpl_lds", "rpl_filter2", "rpl_filter4", "rpl_filter1", "rpl_filter2s", "rpl_filter4s",
                                                      "accel_lbvh", "rpl_cyl1", "rpl_cyl2", "rpl_cyl4", "rpl_cyl2s", "rpl_cyl4s", "rpl_cylw4", "rpl_cylw4s", "rpl_cylm" };

// the two-stage scan variants: paths per lane (R), whether the R paths are consecutive samples of ONE pixel (split) or R
// pixels, and the scan generation (0 = slab filter + LDS queues, sp_filter_scan.h; 1 = cylinder filter + bit words, sp_cyl_scan.h;
// 2 = the same with stage 2 shared by the wave; 3 = stage 1 on the f16 matrix pipe, sp_cylm_scan.h: the default)
struct TwoStage { int R; bool split; int scan; };
bool two_stage(int variant, TwoStage* out) {
	static const TwoStage tab[kVariantLast + 1] = { {0, false, 0}, {0, false, 0}, {0, false, 0}, {2, false, 0}, {4, false, 0}, {1, false, 0}, {2, true, 0}, {4, true, 0},
	                                                {0, false, 0}, {1, false, 1}, {2, false, 1}, {4, false, 1}, {2, true, 1}, {4, true, 1}, {4, false, 2}, {4, true, 2}, {1, false, 3} };
	if (variant < 0 || variant > kVariantLast || tab[variant].R == 0) return false;
	if (out) *out = tab[variant];
	return true;
}

// The shipped library carries the exact-only scans (1, 2), the opt-in BVH (8), one f32 cylinder scan for A/B runs (15) and the
// default (16).  The slab-filter generation and the per-lane cylinder variants are compiled only with -DSP_ALL_VARIANTS (soak and
// experiment builds: tools/).
bool variant_built(int v) {
#ifdef SP_ALL_VARIANTS
	return v >= 1 && v <= kVariantLast;
#else
	return v == 1 || v == 2 || v == kVariantAccel || v == 15 || v == 16;
#endif
}

int pick_variant(int flags, size_t n_tris) {
	if (flags & SPHIP_FLAG_ACCEL) return kVariantAccel;
	const int v = flags & SPHIP_KERNEL_MASK;
	if (v >= 1 && v <= kVariantLast) return v;
	if (n_tris < 64) return 1;        // tiny scenes: nothing to filter, the scalar path has no barriers
	if (n_tris >= (1ull << sp::kMIdxBits)) return 15;      // the default scan packs (ray, triangle index) into 32 bits: 6 + 26
	// The third-generation scan (sp_cylm_scan.h: stage 1 on the f16 matrix pipe, one ray per lane) for every mode; sample chunks
	// (launch_render) supply the workgroups a small frame lacks.
	return 16;
}

// exact records + scene bound: what every variant reads.  The other streams are built on first use (ensure_* below).
int repack(sphip_ctx* c, hipStream_t st) {
	const uint32_t n = (uint32_t)c->n_tris;
	const uint32_t n_pad = (n / sp::kTile + 1) * sp::kTile;      // whole LDS tiles and at least one zero record behind n (index n: the padding of the two-stage streams points at it)
	int rc;
	if ((rc = ensure(c, c->scan, (size_t)n_pad * 48)) || (rc = ensure(c, c->bounds, 256))) return rc;
	HIP_TRY(c, hipMemsetAsync(c->bounds.p, 0, 256, st));
	hipLaunchKernelGGL(sp::k_repack, dim3((n_pad + 255) / 256), dim3(256), 0, st,
	                   (const float*)c->tris.p, (float4*)c->scan.p, (unsigned int*)c->bounds.p, n, n_pad);
	HIP_TRY(c, hipGetLastError());
	c->have_scene = true;
	c->bvh_valid = c->filt_valid = c->cyl_valid = c->cylm_valid = false;
	return SPHIP_OK;
}

// stable radix sort of n (key, value) pairs held in c->sort_kv as keys[2][n], vals[2][n] (sp_radix_sort.h); returns which half holds the result
int radix_sort(sphip_ctx* c, uint32_t n, uint32_t key_bits, hipStream_t st, int* out_half) {
	const uint32_t rs_blocks = (n + sp::kRsPerBlock - 1) / sp::kRsPerBlock;
	int rc = ensure(c, c->sort_hist, (size_t)rs_blocks * 16 * 4);
	if (rc) return rc;
	uint32_t* keys[2] = { (uint32_t*)c->sort_kv.p, (uint32_t*)c->sort_kv.p + (size_t)n };
	uint32_t* vals[2] = { (uint32_t*)c->sort_kv.p + (size_t)2 * n, (uint32_t*)c->sort_kv.p + (size_t)3 * n };
	int cur = 0;
	for (uint32_t shift = 0; shift < key_bits; shift += 4, cur ^= 1) {
		hipLaunchKernelGGL(sp::k_rs_hist, dim3(rs_blocks), dim3(256), 0, st, (const uint32_t*)keys[cur], n, shift, rs_blocks, (uint32_t*)c->sort_hist.p);
		hipLaunchKernelGGL(sp::k_rs_scan, dim3(1), dim3(256), 0, st, (uint32_t*)c->sort_hist.p, rs_blocks * 16u);
		hipLaunchKernelGGL(sp::k_rs_scatter, dim3(rs_blocks), dim3(256), 0, st, (const uint32_t*)keys[cur], (const uint32_t*)vals[cur], n, shift, rs_blocks,
		                   (const uint32_t*)c->sort_hist.p, keys[cur ^ 1], vals[cur ^ 1]);
	}
	HIP_TRY(c, hipGetLastError());
	*out_half = cur;
	return SPHIP_OK;
}

// ---- the default scan's stream (sp_cylm_scan.h): classes by dominant axis, ascending cylinder radius within a class (device sort),
// tiles of kMTile triangles: f32 records + f16 matrix fragments + the per-group Hmax table
template <bool WIDE>
int build_cylm(sphip_ctx* c, hipStream_t st) {
#define SP_CM(x) (WIDE ? sp::cylm512::x : sp::cylm256::x)
	const uint32_t n = (uint32_t)c->n_tris, nblocks = (n + 255) / 256, max_tiles = n / SP_CM(kMTile) + 4;
	int rc;
	if ((rc = ensure(c, c->sort_kv, (size_t)n * 16)) || (rc = ensure(c, c->cylm_hdr, 256)) || (rc = ensure(c, c->cylm_big, sp::cylm256::kMBig * 48)) ||
	    (rc = ensure(c, c->cylm_rec, (size_t)max_tiles * SP_CM(kMTileQ) * 16))) return rc;
	uint32_t* keys = (uint32_t*)c->sort_kv.p;
	uint32_t* vals = (uint32_t*)c->sort_kv.p + (size_t)2 * n;
	uint32_t* hdr = (uint32_t*)c->cylm_hdr.p;
	const float* tris = (const float*)c->tris.p;
	const unsigned int* bnd = (const unsigned int*)c->bounds.p;
	float4* rec = (float4*)c->cylm_rec.p;
	HIP_TRY(c, hipMemsetAsync(hdr, 0, 256, st));
	// (the class, key and big-class kernels do not depend on the tile size)
	hipLaunchKernelGGL(sp::cylm256::k_cylm_count_big, dim3(nblocks), dim3(256), 0, st, tris, n, bnd, hdr);
	hipLaunchKernelGGL(sp::cylm256::k_cylm_keys, dim3(nblocks), dim3(256), 0, st, tris, n, bnd, keys, vals, hdr);
	int half = 0;
	if ((rc = radix_sort(c, n, 32, st, &half))) return rc;
	const uint32_t* sorted = vals + (size_t)half * n;
	if (WIDE) {
		hipLaunchKernelGGL(sp::cylm512::k_cylm_hdr, dim3(1), dim3(1), 0, st, hdr, bnd);
		hipLaunchKernelGGL(sp::cylm512::k_cylm_scatter, dim3(nblocks), dim3(256), 0, st, tris, n, sorted, (const uint32_t*)hdr, rec, (const float4*)c->scan.p, (float4*)c->cylm_big.p);
		hipLaunchKernelGGL(sp::cylm512::k_cylm_pad, dim3(3), dim3(256), 0, st, (const uint32_t*)hdr, n, rec);
		hipLaunchKernelGGL(sp::cylm512::k_cylm_hmax, dim3(max_tiles), dim3(sp::cylm512::kMGroups), 0, st, (const uint32_t*)hdr, rec);
	} else {
		hipLaunchKernelGGL(sp::cylm256::k_cylm_hdr, dim3(1), dim3(1), 0, st, hdr, bnd);
		hipLaunchKernelGGL(sp::cylm256::k_cylm_scatter, dim3(nblocks), dim3(256), 0, st, tris, n, sorted, (const uint32_t*)hdr, rec, (const float4*)c->scan.p, (float4*)c->cylm_big.p);
		hipLaunchKernelGGL(sp::cylm256::k_cylm_pad, dim3(3), dim3(256), 0, st, (const uint32_t*)hdr, n, rec);
		hipLaunchKernelGGL(sp::cylm256::k_cylm_hmax, dim3(max_tiles), dim3(sp::cylm256::kMGroups), 0, st, (const uint32_t*)hdr, rec);
	}
#undef SP_CM
	HIP_TRY(c, hipGetLastError());
	c->cylm_valid = true;
	c->cylm_wide = WIDE;
	return SPHIP_OK;
}

// which shape of the default scan serves this scene (SPATH_HIP_CYLM_SHAPE=256|512 overrides, for A/B runs)
bool cylm_wants_wide(const sphip_ctx* c) {
	if (const char* e = getenv("SPATH_HIP_CYLM_SHAPE")) return atoi(e) == 512;
	return c->n_tris >= sp::kMBigSceneTris;
}

int ensure_cylm(sphip_ctx* c, hipStream_t st) {
	if (c->cylm_valid) return SPHIP_OK;
	return cylm_wants_wide(c) ? build_cylm<true>(c, st) : build_cylm<false>(c, st);
}

// ---- class-sorted f32 cylinder records (sp_cyl_scan.h): count per block -> offsets -> scatter -> pad, all on the device
int ensure_cyl(sphip_ctx* c, hipStream_t st) {
	if (c->cyl_valid) return SPHIP_OK;
	const uint32_t n = (uint32_t)c->n_tris, nblocks = (n + 255) / 256;
	int rc;
	if ((rc = ensure(c, c->cyl_cnt, (size_t)nblocks * 3 * sizeof(uint32_t))) || (rc = ensure(c, c->cyl_hdr, 256)) ||
	    (rc = ensure(c, c->cyl_rec, ((size_t)n / sp::kCylTile + 4) * sp::kCylTile * 32))) return rc;
	hipLaunchKernelGGL(sp::k_cyl_count, dim3(nblocks), dim3(256), 0, st, (const float*)c->tris.p, n, (uint32_t*)c->cyl_cnt.p);
	hipLaunchKernelGGL(sp::k_cyl_offsets, dim3(1), dim3(256), 0, st, (uint32_t*)c->cyl_cnt.p, nblocks, (uint32_t*)c->cyl_hdr.p, sp::kCylTile);
	hipLaunchKernelGGL(sp::k_cyl_scatter, dim3(nblocks), dim3(256), 0, st, (const float*)c->tris.p, n, (const uint32_t*)c->cyl_cnt.p,
	                   (const uint32_t*)c->cyl_hdr.p, (float4*)c->cyl_rec.p);
	hipLaunchKernelGGL(sp::k_cyl_pad, dim3(3), dim3(256), 0, st, (const uint32_t*)c->cyl_hdr.p, n, (float4*)c->cyl_rec.p);
	HIP_TRY(c, hipGetLastError());
	c->cyl_valid = true;
	return SPHIP_OK;
}

#ifdef SP_ALL_VARIANTS
// ---- slab records of the first-generation scan (sp_filter_scan.h)
int ensure_filt(sphip_ctx* c, hipStream_t st) {
	if (c->filt_valid) return SPHIP_OK;
	const uint32_t n = (uint32_t)c->n_tris, n_pad = (n / sp::kTile + 1) * sp::kTile;
	int rc = ensure(c, c->filt, (size_t)n_pad * 48);
	if (rc) return rc;
	hipLaunchKernelGGL(sp::k_repack_filter, dim3((n_pad + 255) / 256), dim3(256), 0, st, (const float*)c->tris.p, (float4*)c->filt.p, n, n_pad);
	HIP_TRY(c, hipGetLastError());
	c->filt_valid = true;
	return SPHIP_OK;
}
#endif

// ---- linear BVH for SPHIP_FLAG_ACCEL (sp_bvh.h), built on the device (sp_bvh_build.h) the first time a scene is rendered with the flag
int ensure_bvh(sphip_ctx* c, hipStream_t st) {
	if (c->bvh_valid) return SPHIP_OK;
	const uint32_t n = (uint32_t)c->n_tris;
	uint32_t nl = 1;
	while ((uint64_t)nl * 4 < n) nl <<= 1;                 // leaves of 4 triangles, padded to a power of two (complete tree in heap order)
	const uint32_t nblocks = (n + 255) / 256;
	int rc;
	if ((rc = ensure(c, c->bvh_nodes, (size_t)2 * nl * 32)) || (rc = ensure(c, c->bvh_rec, ((size_t)nl * 4 + sp::kBvhMaxBig) * 48)) ||
	    (rc = ensure(c, c->bvh_idx, ((size_t)nl * 4 + sp::kBvhMaxBig) * 4)) || (rc = ensure(c, c->sort_kv, (size_t)n * 16)) ||
	    (rc = ensure(c, c->bvh_meta, 256))) return rc;
	uint32_t* meta = (uint32_t*)c->bvh_meta.p;
	uint32_t* vals[2] = { (uint32_t*)c->sort_kv.p + (size_t)2 * n, (uint32_t*)c->sort_kv.p + (size_t)3 * n };
	const float* tris = (const float*)c->tris.p;
	const dim3 b256(256);
	hipLaunchKernelGGL(sp::k_bvh_meta_init, dim3(1), b256, 0, st, meta);
	hipLaunchKernelGGL(sp::k_bvh_box, dim3(nblocks), b256, 0, st, tris, n, meta);
	hipLaunchKernelGGL(sp::k_bvh_count_big, dim3(nblocks), b256, 0, st, tris, n, meta);
	hipLaunchKernelGGL(sp::k_bvh_keys, dim3(nblocks), b256, 0, st, tris, n, meta, (uint32_t*)c->sort_kv.p, vals[0]);
	int cur = 0;
	if ((rc = radix_sort(c, n, 32, st, &cur))) return rc;            // stable LSD radix sort by (Morton code; big triangles last)
	hipLaunchKernelGGL(sp::k_bvh_leaves, dim3((nl + 255) / 256), b256, 0, st, tris, n, (const uint32_t*)meta, (const uint32_t*)vals[cur], nl,
	                   (float4*)c->bvh_nodes.p, (float4*)c->bvh_rec.p, (int*)c->bvh_idx.p);
	hipLaunchKernelGGL(sp::k_bvh_bigs, dim3(1), b256, 0, st, tris, n, (const uint32_t*)meta, (const uint32_t*)vals[cur], nl, (float4*)c->bvh_rec.p, (int*)c->bvh_idx.p);
	for (uint32_t first = nl >> 1; first >= 1; first >>= 1)              // bottom-up, one level per launch
		hipLaunchKernelGGL(sp::k_bvh_refit, dim3((first + 255) / 256), b256, 0, st, (float4*)c->bvh_nodes.p, first);
	HIP_TRY(c, hipGetLastError());
	c->bvh_leaves = nl;
	c->bvh_valid = true;
	return SPHIP_OK;
}

constexpr int kModeHits = 2;   // internal: sphip_closest_hit_device

int launch_render(sphip_ctx* c, const void* d_rays, size_t n_rays, const sphip_shard* shard, size_t /*image_width*/,
                  size_t n_samples, uint64_t seed, int mode, int flags, void* d_rgba, void* d_accum, hipStream_t st,
                  const int* d_src = nullptr) {
	if (!c->have_scene) return fail(c, SPHIP_E_STATE, "render called before a scene was set");
	if (!d_rays || !d_rgba) return fail(c, SPHIP_E_INVALID, "null ray or output pointer");
	if (n_rays == 0 || n_rays > 0xffffffffull) return fail(c, SPHIP_E_INVALID, "n_rays %zu out of range", n_rays);
	if (mode != SPHIP_MODE_FLAT && mode != SPHIP_MODE_PT && mode != kModeHits) return fail(c, SPHIP_E_INVALID, "unknown mode %d", mode);
	if (mode == kModeHits && !d_accum) return fail(c, SPHIP_E_INVALID, "null distance output");
	if (mode == SPHIP_MODE_PT && (n_samples == 0 || n_samples > 0x7fffffffull))
		return fail(c, SPHIP_E_INVALID, "n_samples must be in [1, 2^31) (the reference divides by it, cpu_renderer.cpp:77)");
	int rc = ensure(c, c->counter, 16 * sizeof(unsigned long long));
	if (rc) return rc;

	sp::KArgs a{};
	a.rays = (const float*)d_rays;
	a.scan = (const float4*)c->scan.p;
	a.tris = (const float*)c->tris.p;
	a.mats = (const float*)c->mats.p;
	a.out_rgba = (uint32_t*)d_rgba;
	a.out_accum = (float*)d_accum;
	a.scans = (unsigned long long*)c->counter.p;
	a.n_rays = (uint32_t)n_rays;
	a.n_tris = (uint32_t)c->n_tris;
	a.n_samples = (uint32_t)n_samples;
	a.flags = (uint32_t)flags;
	a.seed = seed;
	if (shard) {
		if (shard->tile_px == 0) return fail(c, SPHIP_E_INVALID, "shard.tile_px must be > 0");
		a.pixel_base = shard->pixel_base; a.tile_px = shard->tile_px; a.tile_stride_px = shard->tile_stride_px;
	} else {
		a.pixel_base = 0; a.tile_px = n_rays; a.tile_stride_px = 0;
	}
	a.inv_n = (float)(1.0 / (double)(n_samples ? n_samples : 1));      // cpu_renderer.cpp:77

	const int variant = pick_variant(flags, c->n_tris);
	if (variant == 16 && c->n_tris >= (1ull << sp::kMIdxBits))
		return fail(c, SPHIP_E_INVALID, "rpl_cylm handles scenes of fewer than 2^%u triangles (this one has %zu); use rpl_cylw4s", sp::kMIdxBits, c->n_tris);
	if (!variant_built(variant))
		return fail(c, SPHIP_E_INVALID, "kernel variant %d (%s) is not compiled into this build of libspath_hip (rebuild with -DSP_ALL_VARIANTS)", variant, kVariantNames[variant]);
	HIP_TRY(c, hipMemsetAsync(c->counter.p, 0, 16 * sizeof(unsigned long long), st));
	// sample chunks: the filter kernels keep 1024 workgroups resident (256 CUs x 4); a launch of only a few times that
	// many ends with a long tail (its time is that of the slowest workgroup, ~12 % above the mean when everything starts
	// together), so small frames and multi-GPU shards are split along the samples as well
	uint32_t chunks = 1;
	TwoStage ts{1, false, 0};
	const bool is_ts = two_stage(variant, &ts);
	const uint32_t slots = is_ts && ts.split ? (uint32_t)ts.R : 1u;                       // samples of one pixel per lane
	// the record stream the variant reads, derived from the scene on first use
	if (is_ts && ts.scan == 3 && (rc = ensure_cylm(c, st))) return rc;
	if (is_ts && ts.scan == 3 && c->cylm_wide) ts.scan = 4;                               // the 512-thread shape of the same scan (sp_cylm_both.h)
	const uint32_t bthreads = is_ts && ts.scan == 4 ? sp::cylm512::kMThreads : 256u;      // threads per workgroup of the variant's kernels
	const uint32_t rays_per_block = is_ts && !ts.split ? bthreads * (uint32_t)ts.R : bthreads;
	const uint64_t chunk_target = kChunkTargetBlocks * 256u / bthreads;                   // the same number of resident-wave rounds
	if (mode == SPHIP_MODE_PT && is_ts) {
		const uint64_t px_blocks = (n_rays + rays_per_block - 1) / rays_per_block;
		const uint64_t n_iter = (n_samples + slots - 1) / slots;
		const uint32_t forced = ((uint32_t)flags & SPHIP_FLAG_CHUNKS_MASK) >> SPHIP_FLAG_CHUNKS_SHIFT;
		if (forced) chunks = forced;
		else while (px_blocks * chunks < chunk_target && chunks < 128) chunks *= 2;
		if (chunks > n_iter) chunks = (uint32_t)n_iter;
		const uint64_t samp_bytes = (uint64_t)n_samples * 12 * ((n_rays + 255) / 256 * 256);
		if (chunks > 1 && !forced) {
			// the split is an optimisation: it must never make a render fail that would fit unsplit (52 B per pixel).
			// Keep the scratch under the cap and under 90 % of what the device has free beyond the cached buffers.
			// (the device is asked for its free memory only when a buffer has to grow: not on every frame of a running viewer)
			size_t free_b = 0, total_b = 0;
			bool asked = false;
			if (samp_bytes > kChunkMaxBytes) chunks = 1;
			const uint64_t lanes = (uint64_t)((n_rays + 1023) / 1024 * 1024) * slots;
			while (chunks > 1) {
				const uint64_t work_bytes = lanes * chunks * 52;
				const uint64_t need = (samp_bytes > c->samp.cap ? samp_bytes : 0) + (work_bytes > c->work.cap ? work_bytes : 0);
				if (need == 0) break;
				if (!asked) { asked = true; if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { chunks = 1; break; } }
				if (need <= (uint64_t)((double)free_b * 0.9)) break;
				chunks /= 2;
			}
		}
		if (chunks > 1) {
			a.n_chunks = chunks; a.px_blocks = (uint32_t)px_blocks; a.samp_stride = (uint32_t)((n_rays + 255) / 256 * 256);
			if ((rc = ensure(c, c->samp, (size_t)samp_bytes))) return rc;
			a.samp = (float*)c->samp.p;
		}
	}
	const dim3 block(256), block_ts(bthreads);
	const dim3 grid((unsigned)((n_rays + 255) / 256 * chunks));                           // exact-only kernels, accel
	// one-scan modes (flat pass, hits) have no samples to share a lane: a split variant runs there with R pixels per lane
	const uint32_t rpb1 = is_ts ? bthreads * (uint32_t)ts.R : 256u;
	const dim3 grid_px((unsigned)((n_rays + rpb1 - 1) / rpb1));
	const dim3 grid_pt((unsigned)((n_rays + rays_per_block - 1) / rays_per_block * chunks));
	// path-history / accumulator work buffer of the two-stage kernels: 5 x int2 + 3 x float per (padded) path and chunk
	const uint64_t n_work64 = (uint64_t)((n_rays + 1023) / 1024 * 1024) * slots * chunks;
	if (mode == SPHIP_MODE_PT && is_ts && n_work64 > 0xffffffffull)
		return fail(c, SPHIP_E_INVALID, "n_rays %zu too large for one launch of this kernel variant; shard the frame", n_rays);
	const uint32_t n_work = (uint32_t)n_work64;
	int2* hist = nullptr; float* acc = nullptr;
	if (mode == SPHIP_MODE_PT && is_ts) {
		if ((rc = ensure(c, c->work, (size_t)n_work * 52))) return rc;
		hist = (int2*)c->work.p;
		acc = (float*)((char*)c->work.p + (size_t)n_work * 40);
	}
	sp::ScanSrc src2{};
	if (is_ts && (ts.scan == 1 || ts.scan == 2) && (rc = ensure_cyl(c, st))) return rc;
#ifdef SP_ALL_VARIANTS
	if (is_ts && ts.scan == 0 && (rc = ensure_filt(c, st))) return rc;
#endif
	src2.filt = (const float4*)c->filt.p;
	src2.cyl.rec = (const float4*)c->cyl_rec.p;
	src2.cyl.hdr = (const uint32_t*)c->cyl_hdr.p;
	src2.cylm.rec = (const float4*)c->cylm_rec.p;
	src2.cylm.hdr = (const uint32_t*)c->cylm_hdr.p;
	src2.cylm.big = (const float4*)c->cylm_big.p;
	const unsigned int* bnd = (const unsigned int*)c->bounds.p;
	sp::BvhArgs B{};
	if (variant == kVariantAccel) {
		if ((rc = ensure_bvh(c, st))) return rc;
		B.nodes = (const float4*)c->bvh_nodes.p; B.leaf_rec = (const float4*)c->bvh_rec.p; B.leaf_idx = (const int*)c->bvh_idx.p;
		B.n_leaves = c->bvh_leaves; B.first_leaf = c->bvh_leaves; B.meta = (const uint32_t*)c->bvh_meta.p;
	}
	HIP_TRY(c, hipEventRecord(c->ev_k0, st));
	// primary-hit reuse with a two-stage kernel: one closest-hit scan per PIXEL first (R pixels per lane, the same scan family),
	// whatever the number of samples and sample chunks; the path-tracing launch then starts every sample from that hit
	bool prim_pass = false;
	if (mode == SPHIP_MODE_PT && is_ts && (flags & SPHIP_FLAG_PRIMARY_REUSE)) {
		if ((rc = ensure(c, c->prim, n_rays * 8))) return rc;
		int* oi = (int*)c->prim.p; float* od = (float*)((char*)c->prim.p + n_rays * 4);
		sp::KArgs h = a;
		h.n_chunks = 0; h.samp = nullptr;
		const int* no_src = nullptr;
#define SP_PRIM(R_, S_) hipLaunchKernelGGL((sp::k_hit_filter<R_, S_>), grid_px, block_ts, 0, st, h, src2, bnd, no_src, oi, od)
		if (ts.scan == 3) SP_PRIM(1, 3);
		else if (ts.scan == 4) SP_PRIM(1, 4);
		else if (ts.scan == 2) SP_PRIM(4, 2);
#ifdef SP_ALL_VARIANTS
		else if (ts.scan == 0) { if (ts.R == 4) SP_PRIM(4, 0); else if (ts.R == 2) SP_PRIM(2, 0); else SP_PRIM(1, 0); }
		else              { if (ts.R == 4) SP_PRIM(4, 1); else if (ts.R == 2) SP_PRIM(2, 1); else SP_PRIM(1, 1); }
#endif
#undef SP_PRIM
		a.prim_idx = oi; a.prim_d = od;
		prim_pass = true;
	}
	if (variant == kVariantAccel) {
		if (mode == kModeHits)           hipLaunchKernelGGL(sp::k_accel<2>, grid, block, 0, st, a, B, d_src, (int*)d_rgba, (float*)d_accum);
		else if (mode == SPHIP_MODE_FLAT) hipLaunchKernelGGL(sp::k_accel<0>, grid, block, 0, st, a, B, nullptr, nullptr, nullptr);
		else                              hipLaunchKernelGGL(sp::k_accel<1>, grid, block, 0, st, a, B, nullptr, nullptr, nullptr);
	} else if (mode == kModeHits) {
		int* oi = (int*)d_rgba; float* od = (float*)d_accum;
		if (is_ts) {
#define SP_HIT(R_, S_) hipLaunchKernelGGL((sp::k_hit_filter<R_, S_>), grid_px, block_ts, 0, st, a, src2, bnd, d_src, oi, od)
			if (ts.scan == 3) SP_HIT(1, 3);
			else if (ts.scan == 4) SP_HIT(1, 4);
			else if (ts.scan == 2) SP_HIT(4, 2);
#ifdef SP_ALL_VARIANTS
			else if (ts.scan == 0) { if (ts.R == 4) SP_HIT(4, 0); else if (ts.R == 2) SP_HIT(2, 0); else SP_HIT(1, 0); }
			else              { if (ts.R == 4) SP_HIT(4, 1); else if (ts.R == 2) SP_HIT(2, 1); else SP_HIT(1, 1); }
#endif
#undef SP_HIT
		}
		else if (variant == 2) hipLaunchKernelGGL(sp::k_hit<2>, grid, block, 0, st, a, d_src, oi, od);
		else                   hipLaunchKernelGGL(sp::k_hit<1>, grid, block, 0, st, a, d_src, oi, od);
	} else if (mode == SPHIP_MODE_FLAT) {
		if (is_ts) {
#define SP_FLAT(R_, S_) hipLaunchKernelGGL((sp::k_flat_filter<R_, S_>), grid_px, block_ts, 0, st, a, src2, bnd)
			if (ts.scan == 3) SP_FLAT(1, 3);
			else if (ts.scan == 4) SP_FLAT(1, 4);
			else if (ts.scan == 2) SP_FLAT(4, 2);
#ifdef SP_ALL_VARIANTS
			else if (ts.scan == 0) { if (ts.R == 4) SP_FLAT(4, 0); else if (ts.R == 2) SP_FLAT(2, 0); else SP_FLAT(1, 0); }
			else              { if (ts.R == 4) SP_FLAT(4, 1); else if (ts.R == 2) SP_FLAT(2, 1); else SP_FLAT(1, 1); }
#endif
#undef SP_FLAT
		}
		else if (variant == 2) hipLaunchKernelGGL(sp::k_flat<2>, grid, block, 0, st, a);
		else                   hipLaunchKernelGGL(sp::k_flat<1>, grid, block, 0, st, a);
	} else {
		if (is_ts) {
#define SP_PT(R_, SPLIT_, S_) hipLaunchKernelGGL((sp::k_pt_filter<R_, SPLIT_, S_>), grid_pt, block_ts, 0, st, a, src2, bnd, hist, acc, n_work)
			if (ts.scan == 3) SP_PT(1, false, 3);
			else if (ts.scan == 4) SP_PT(1, false, 4);
			else if (ts.scan == 2 && ts.split) SP_PT(4, true, 2);
#ifdef SP_ALL_VARIANTS
			else if (ts.scan == 2) SP_PT(4, false, 2);
			else if (ts.scan == 0) {
				if (ts.split) { if (ts.R == 4) SP_PT(4, true, 0); else SP_PT(2, true, 0); }
				else          { if (ts.R == 4) SP_PT(4, false, 0); else if (ts.R == 2) SP_PT(2, false, 0); else SP_PT(1, false, 0); }
			} else {
				if (ts.split) { if (ts.R == 4) SP_PT(4, true, 1); else SP_PT(2, true, 1); }
				else          { if (ts.R == 4) SP_PT(4, false, 1); else if (ts.R == 2) SP_PT(2, false, 1); else SP_PT(1, false, 1); }
			}
#endif
#undef SP_PT
		}
		else if (variant == 2) hipLaunchKernelGGL(sp::k_pt<2>, grid, block, 0, st, a);
		else                   hipLaunchKernelGGL(sp::k_pt<1>, grid, block, 0, st, a);
	}
	if (chunks > 1) hipLaunchKernelGGL(sp::k_resolve, dim3((unsigned)((n_rays + 255) / 256)), block, 0, st, a);
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipEventRecord(c->ev_k1, st));
	c->have_render = true;
	c->last_stream = st;
	c->stats.n_tris = c->n_tris;
	c->stats.n_pixels = n_rays;
	c->stats.kernel_variant = (uint32_t)variant;
	c->stats.n_launches = (chunks > 1 ? 2u : 1u) + (prim_pass ? 1u : 0u);
	return SPHIP_OK;
}


// ---- view::camera::get_viewport on the device for the pixels of a shard (sp_kernels.h: k_viewport)
int launch_viewport(sphip_ctx* c, const sphip_camera* cam, void* d_rays, hipStream_t st, const sphip_shard* shard = nullptr, size_t n_local = 0) {
	if (!cam || !d_rays) return fail(c, SPHIP_E_INVALID, "null camera or ray pointer");
	if (cam->res_x == 0 || cam->res_y == 0 || (uint64_t)cam->res_x * cam->res_y > 0xffffffffull)
		return fail(c, SPHIP_E_INVALID, "bad viewport size %ux%u", cam->res_x, cam->res_y);
	sp::ViewArgs v{};
	// view.h:101-108: `real` (float) variables initialised from double expressions
	const float x_size = (float)(1.0 * (double)cam->res_x / (double)cam->res_y), y_size = 1.0f;
	v.x_max = (float)((double)x_size / 2.0);
	v.x_step = x_size / (float)cam->res_x;
	v.h_x_step = (float)((double)v.x_step / 2.0);
	v.y_max = (float)((double)y_size / 2.0);
	v.y_step = y_size / (float)cam->res_y;
	v.h_y_step = (float)((double)v.y_step / 2.0);
	v.focal = cam->focal; v.cos_y = cam->cos_y; v.sin_y = cam->sin_y; v.cos_x = cam->cos_x; v.sin_x = cam->sin_x;
	v.px = cam->pos[0]; v.py = cam->pos[1]; v.pz = cam->pos[2];
	v.res_x = cam->res_x; v.res_y = cam->res_y;
	const uint32_t n = shard ? (uint32_t)n_local : cam->res_x * cam->res_y;
	v.n_local = n;
	if (shard) { v.pixel_base = shard->pixel_base; v.tile_px = shard->tile_px; v.tile_stride_px = shard->tile_stride_px; }
	else { v.pixel_base = 0; v.tile_px = n; v.tile_stride_px = 0; }
	if (n == 0) return SPHIP_OK;
	hipLaunchKernelGGL(sp::k_viewport, dim3((n + 255) / 256), dim3(256), 0, st, v, (float*)d_rays);
	HIP_TRY(c, hipGetLastError());
	return SPHIP_OK;
}


// =====================================================================================================================
// All GPUs of a node behind one context (include/spath_hip.h: sphip_create_multi).  The reference has no multi-device
// code; what is sharded is the pixel loop of cpu_renderer.cpp:70-79,118-184 (pixels are independent), behind the same
// renderer::render / render_flat calls (src/renderer.h:31-32).
// =====================================================================================================================

// ---- the row-tile plan
int plan_tile_rows(size_t height, int n_dev) {
	for (int tr = 8; tr >= 1; --tr)
		if (height % (size_t)tr == 0 && (height / (size_t)tr) % (size_t)n_dev == 0) return tr;
	return 8;
}

struct RowPlan {
	size_t w, h, tile_rows, tile_px, n_tiles, npix;
	int g;
	RowPlan(size_t w_, size_t h_, int g_, size_t tr) : w(w_), h(h_), tile_rows(tr), tile_px(tr * w_), n_tiles((h_ + tr - 1) / tr), npix(w_ * h_), g(g_) {}
	size_t n_rays(int rank) const {
		size_t n = 0;
		for (size_t t = (size_t)rank; t < n_tiles; t += (size_t)g) n += std::min(tile_px, npix - t * tile_px);
		return n;
	}
	size_t max_rays() const { size_t m = 0; for (int r = 0; r < g; ++r) m = std::max(m, n_rays(r)); return m; }
	sphip_shard shard(int rank) const { return sphip_shard{ (uint64_t)rank * tile_px, (uint64_t)tile_px, (uint64_t)g * tile_px }; }
};

// ---- RCCL, loaded on first use: the single-GPU path (and every process that never creates a multi-device context) does not
// depend on librccl being present
typedef int (*nccl_init_all_t)(void**, int, const int*);
typedef int (*nccl_comm_destroy_t)(void*);
typedef int (*nccl_group_t)(void);
typedef int (*nccl_send_t)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_recv_t)(void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_t)(int);
struct Rccl {
	nccl_init_all_t init_all = nullptr; nccl_comm_destroy_t destroy = nullptr; nccl_group_t group_start = nullptr, group_end = nullptr;
	nccl_send_t send = nullptr; nccl_recv_t recv = nullptr; nccl_errstr_t errstr = nullptr;
} g_rccl;
constexpr int kNcclUint8 = 1;      // ncclDataType_t::ncclUint8 (rccl.h)

// (once per process, under a lock: contexts may be created from several threads; the table is never changed afterwards and the
// library stays loaded)
void* load_rccl() {
	static std::mutex mu;
	static void* loaded = nullptr;
	std::lock_guard<std::mutex> lock(mu);
	if (loaded) return loaded;
	const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	for (const char* n : names) {
		void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
		if (!h) continue;
		g_rccl.init_all = (nccl_init_all_t)dlsym(h, "ncclCommInitAll");
		g_rccl.destroy = (nccl_comm_destroy_t)dlsym(h, "ncclCommDestroy");
		g_rccl.group_start = (nccl_group_t)dlsym(h, "ncclGroupStart");
		g_rccl.group_end = (nccl_group_t)dlsym(h, "ncclGroupEnd");
		g_rccl.send = (nccl_send_t)dlsym(h, "ncclSend");
		g_rccl.recv = (nccl_recv_t)dlsym(h, "ncclRecv");
		g_rccl.errstr = (nccl_errstr_t)dlsym(h, "ncclGetErrorString");
		if (g_rccl.init_all && g_rccl.destroy && g_rccl.group_start && g_rccl.group_end && g_rccl.send && g_rccl.recv) { loaded = h; return h; }
		g_rccl = Rccl{};
		dlclose(h);
	}
	return nullptr;
}

int multi_set_scene(sphip_ctx* c, const float* tris, const float* mats, size_t n_tris) {
	const int g = (int)c->kids.size();
	std::vector<int> rcs((size_t)g, SPHIP_OK);
	std::vector<std::thread> th;
	for (int r = 0; r < g; ++r) th.emplace_back([&, r] { rcs[(size_t)r] = sphip_set_scene(c->kids[(size_t)r], tris, mats, n_tris); });   // every device holds the whole scene
	for (auto& t : th) t.join();
	for (int r = 0; r < g; ++r)
		if (rcs[(size_t)r]) return fail(c, rcs[(size_t)r], "device %d: %s", c->kids[(size_t)r]->device, c->kids[(size_t)r]->err.c_str());
	c->n_tris = n_tris;
	c->have_scene = true;
	return SPHIP_OK;
}

int multi_render_impl(sphip_ctx* c, const float* rays, const sphip_camera* cam, size_t w, size_t h, size_t n_samples, uint64_t seed, int mode, int flags,
                      uint8_t* out_rgba, float* out_accum);

// rays != nullptr: the caller's viewport (host array, w*h rays); else cam: every device generates the rays of its own tiles.
// Whatever goes wrong on one device, no device is left with work in flight when the call returns: the next call may free or
// regrow the buffers that work reads and writes.
int multi_render(sphip_ctx* c, const float* rays, const sphip_camera* cam, size_t w, size_t h, size_t n_samples, uint64_t seed, int mode, int flags,
                 uint8_t* out_rgba, float* out_accum) {
	const int rc = multi_render_impl(c, rays, cam, w, h, n_samples, seed, mode, flags, out_rgba, out_accum);
	if (rc != SPHIP_OK) {
		for (sphip_ctx* k : c->kids)
			if (hipSetDevice(k->device) == hipSuccess && k->own_stream) (void)hipStreamSynchronize(k->own_stream);
		(void)hipGetLastError();
	}
	return rc;
}

int multi_render_impl(sphip_ctx* c, const float* rays, const sphip_camera* cam, size_t w, size_t h, size_t n_samples, uint64_t seed, int mode, int flags,
                      uint8_t* out_rgba, float* out_accum) {
	if (!c->have_scene) return fail(c, SPHIP_E_STATE, "render called before a scene was set");
	if (!out_rgba || w == 0 || h == 0 || w * h > 0xffffffffull) return fail(c, SPHIP_E_INVALID, "bad render arguments (w=%zu h=%zu)", w, h);
	if (mode == SPHIP_MODE_PT && (n_samples == 0 || n_samples > 0x7fffffffull))
		return fail(c, SPHIP_E_INVALID, "n_samples must be in [1, 2^31) (the reference divides by it, cpu_renderer.cpp:77)");
	const int g = (int)c->kids.size();
	const RowPlan plan(w, h, g, (size_t)plan_tile_rows(h, g));
	const size_t pad = plan.max_rays(), npix = plan.npix;
	sphip_ctx* root = c->kids[0];
	HIP_TRY(c, hipSetDevice(root->device));
	int rc;
	if ((rc = ensure(c, c->gath, (size_t)g * pad * 4)) || (rc = ensure(c, c->img, npix * 4))) return rc;
	if (out_accum && ((rc = ensure(c, c->gath_acc, (size_t)g * pad * 12)) || (rc = ensure(c, c->img_acc, npix * 12)))) return rc;
	HIP_TRY(c, hipEventRecord(c->ev_g0, root->own_stream));
	const bool peer = c->gather_kind != SPHIP_GATHER_RCCL;
	std::vector<int> rcs((size_t)g, SPHIP_OK);
	std::vector<std::thread> th;
	for (int r = 0; r < g; ++r) th.emplace_back([&, r] {
		sphip_ctx* k = c->kids[(size_t)r];
		auto body = [&]() -> int {
			const size_t n = plan.n_rays(r);
			k->have_render = false;
			if (n == 0) return SPHIP_OK;
			HIP_TRY(k, hipSetDevice(k->device));
			hipStream_t st = k->own_stream;
			int rc2;
			if ((rc2 = ensure(k, k->rays, n * 24)) || (rc2 = ensure(k, k->rgba, pad * 4))) return rc2;
			if (out_accum && (rc2 = ensure(k, k->accum, pad * 12))) return rc2;
			const sphip_shard sh = plan.shard(r);
			if (rays) {           // this device's tiles of the caller's viewport, one copy per tile
				size_t k0 = 0;
				for (size_t t = (size_t)r; t < plan.n_tiles; t += (size_t)g) {
					const size_t cnt = std::min(plan.tile_px, npix - t * plan.tile_px);
					HIP_TRY(k, hipMemcpyAsync((char*)k->rays.p + k0 * 24, rays + t * plan.tile_px * 6, cnt * 24, hipMemcpyHostToDevice, st));
					k0 += cnt;
				}
			} else if ((rc2 = launch_viewport(k, cam, k->rays.p, st, &sh, n))) return rc2;
			if ((rc2 = launch_render(k, k->rays.p, n, &sh, w, n_samples, seed, mode, flags, k->rgba.p, out_accum ? k->accum.p : nullptr, st))) return rc2;
			k->timed_upload = k->timed_download = false;
			if (peer) {           // tiles -> slot r of the first device's gather buffer, in stream order behind the kernels
				const bool same = k->device == root->device;      // a device listed twice, or the first device itself: a local copy
				if (same) HIP_TRY(k, hipMemcpyAsync((char*)c->gath.p + (size_t)r * pad * 4, k->rgba.p, n * 4, hipMemcpyDeviceToDevice, st));
				else HIP_TRY(k, hipMemcpyPeerAsync((char*)c->gath.p + (size_t)r * pad * 4, root->device, k->rgba.p, k->device, n * 4, st));
				if (out_accum && same) HIP_TRY(k, hipMemcpyAsync((char*)c->gath_acc.p + (size_t)r * pad * 12, k->accum.p, n * 12, hipMemcpyDeviceToDevice, st));
				else if (out_accum) HIP_TRY(k, hipMemcpyPeerAsync((char*)c->gath_acc.p + (size_t)r * pad * 12, root->device, k->accum.p, k->device, n * 12, st));
				HIP_TRY(k, hipEventRecord(c->ev_tile[(size_t)r], st));
			}
			return SPHIP_OK;
		};
		rcs[(size_t)r] = body();
	});
	for (auto& t : th) t.join();
	for (int r = 0; r < g; ++r)
		if (rcs[(size_t)r]) return fail(c, rcs[(size_t)r], "device %d: %s", c->kids[(size_t)r]->device, c->kids[(size_t)r]->err.c_str());
	HIP_TRY(c, hipSetDevice(root->device));
	hipStream_t rs = root->own_stream;
	// peer-copy exchange, issued from this thread: device r's tiles -> slot r of the first device's buffer, in stream order behind its
	// kernels (the host threads do the same themselves when peer copies are the context's exchange from the start)
	auto gather_peer_now = [&]() -> int {
		for (int r = 0; r < g; ++r) {
			const size_t n = plan.n_rays(r);
			if (!n) continue;
			sphip_ctx* k = c->kids[(size_t)r];
			HIP_TRY(c, hipSetDevice(k->device));
			const bool same = k->device == root->device;
			if (same) HIP_TRY(c, hipMemcpyAsync((char*)c->gath.p + (size_t)r * pad * 4, k->rgba.p, n * 4, hipMemcpyDeviceToDevice, k->own_stream));
			else HIP_TRY(c, hipMemcpyPeerAsync((char*)c->gath.p + (size_t)r * pad * 4, root->device, k->rgba.p, k->device, n * 4, k->own_stream));
			if (out_accum && same) HIP_TRY(c, hipMemcpyAsync((char*)c->gath_acc.p + (size_t)r * pad * 12, k->accum.p, n * 12, hipMemcpyDeviceToDevice, k->own_stream));
			else if (out_accum) HIP_TRY(c, hipMemcpyPeerAsync((char*)c->gath_acc.p + (size_t)r * pad * 12, root->device, k->accum.p, k->device, n * 12, k->own_stream));
			HIP_TRY(c, hipEventRecord(c->ev_tile[(size_t)r], k->own_stream));
		}
		HIP_TRY(c, hipSetDevice(root->device));
		return SPHIP_OK;
	};
	bool wait_tiles = peer;
	if (!peer) {
		// one grouped exchange: every other device sends its tiles, the first device receives them into its gather buffer;
		// its own tiles are a local copy
		if (plan.n_rays(0)) {
			HIP_TRY(c, hipMemcpyAsync(c->gath.p, root->rgba.p, plan.n_rays(0) * 4, hipMemcpyDeviceToDevice, rs));
			if (out_accum) HIP_TRY(c, hipMemcpyAsync(c->gath_acc.p, root->accum.p, plan.n_rays(0) * 12, hipMemcpyDeviceToDevice, rs));
		}
		int nrc = g_rccl.group_start();
		for (int r = 1; r < g && !nrc; ++r) {
			const size_t n = plan.n_rays(r);
			if (!n) continue;
			sphip_ctx* k = c->kids[(size_t)r];
			if (!nrc) nrc = g_rccl.send(k->rgba.p, n * 4, kNcclUint8, 0, c->comms[(size_t)r], k->own_stream);
			if (!nrc) nrc = g_rccl.recv((char*)c->gath.p + (size_t)r * pad * 4, n * 4, kNcclUint8, r, c->comms[0], rs);
			if (out_accum && !nrc) nrc = g_rccl.send(k->accum.p, n * 12, kNcclUint8, 0, c->comms[(size_t)r], k->own_stream);
			if (out_accum && !nrc) nrc = g_rccl.recv((char*)c->gath_acc.p + (size_t)r * pad * 12, n * 12, kNcclUint8, r, c->comms[0], rs);
		}
		const int erc = g_rccl.group_end();
		if (nrc || erc) {
			// the communicator did not take the exchange: this frame and every later one go through peer copies (the kernels'
			// results are still in each device's buffer); say so once, loudly
			fprintf(stderr, "libspath_hip: RCCL gather failed (%s); falling back to peer copies\n", g_rccl.errstr ? g_rccl.errstr(nrc ? nrc : erc) : "?");
			c->gather_kind = SPHIP_GATHER_PEER;
			if ((rc = gather_peer_now())) return rc;
			wait_tiles = true;
		}
	}
	if (wait_tiles)
		for (int r = 1; r < g; ++r) if (plan.n_rays(r)) HIP_TRY(c, hipStreamWaitEvent(rs, c->ev_tile[(size_t)r], 0));
	const dim3 grid((unsigned)((npix + 255) / 256)), block(256);
	hipLaunchKernelGGL(sp::k_assemble<1>, grid, block, 0, rs, (const uint32_t*)c->gath.p, (uint32_t*)c->img.p, (uint32_t)npix, (uint32_t)plan.tile_px, (uint32_t)g, (uint32_t)pad);
	if (out_accum)
		hipLaunchKernelGGL(sp::k_assemble<3>, grid, block, 0, rs, (const uint32_t*)c->gath_acc.p, (uint32_t*)c->img_acc.p, (uint32_t)npix, (uint32_t)plan.tile_px, (uint32_t)g, (uint32_t)pad);
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipEventRecord(c->ev_g1, rs));
	HIP_TRY(c, hipMemcpyAsync(out_rgba, c->img.p, npix * 4, hipMemcpyDeviceToHost, rs));
	if (out_accum) HIP_TRY(c, hipMemcpyAsync(out_accum, c->img_acc.p, npix * 12, hipMemcpyDeviceToHost, rs));
	HIP_TRY(c, hipStreamSynchronize(rs));            // blocking, like every reference backend (main.cpp:70-83)
	for (int r = 1; r < g; ++r) {                    // the other devices' streams are idle now too (their last work fed the gather)
		HIP_TRY(c, hipSetDevice(c->kids[(size_t)r]->device));
		HIP_TRY(c, hipStreamSynchronize(c->kids[(size_t)r]->own_stream));
	}
	c->have_render = true;
	c->stats.n_pixels = npix;
	c->stats.n_tris = c->n_tris;
	return SPHIP_OK;
}

int multi_get_stats(sphip_ctx* c, sphip_stats* out) {
	if (!c->have_render) return fail(c, SPHIP_E_STATE, "no render has been issued yet");
	sphip_stats s{};
	s.kernel_ms_min = 1e300;
	for (sphip_ctx* k : c->kids) {
		if (!k->have_render) continue;
		sphip_stats ks;
		const int rc = sphip_get_stats(k, &ks);
		if (rc) return fail(c, rc, "device %d: %s", k->device, k->err.c_str());
		s.kernel_ms = std::max(s.kernel_ms, ks.kernel_ms);
		s.kernel_ms_min = std::min(s.kernel_ms_min, ks.kernel_ms);
		s.scans_executed += ks.scans_executed;
		s.kernel_variant = ks.kernel_variant;
		s.n_launches = ks.n_launches;
		++s.n_devices;
	}
	HIP_TRY(c, hipSetDevice(c->kids[0]->device));
	float ms = 0.0f;
	HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_g0, c->ev_g1));   // includes the first device's own kernels: report what is beyond them
	s.gather_ms = std::max(0.0, (double)ms - s.kernel_ms);
	s.gather_kind = (uint32_t)c->gather_kind;
	s.n_tris = c->n_tris;
	s.n_pixels = c->stats.n_pixels;
	c->stats = s;
	*out = s;
	return SPHIP_OK;
}

} // namespace

extern "C" {

int sphip_abi_version(void) { return SPHIP_ABI_VERSION; }

const char* sphip_kernel_name(int variant) {
	if (variant < 0 || variant > kVariantLast) return nullptr;
	return kVariantNames[variant];
}

#ifndef SP_SOURCE_HASH
#define SP_SOURCE_HASH "unknown"
#endif
const char* sphip_build_info(void) { return "src=" SP_SOURCE_HASH; }

int sphip_kernel_available(int variant) { return variant_built(variant) ? 1 : 0; }

int sphip_plan_tile_rows(size_t height, int n_devices) {
	if (height == 0 || n_devices < 1) return SPHIP_E_INVALID;
	return plan_tile_rows(height, n_devices);
}

int sphip_plan_shard(size_t width, size_t height, int n_devices, size_t tile_rows, int rank, sphip_shard* shard_out, size_t* n_rays_out) {
	if (width == 0 || height == 0 || n_devices < 1 || tile_rows == 0 || rank < 0 || rank >= n_devices) return SPHIP_E_INVALID;
	const RowPlan plan(width, height, n_devices, tile_rows);
	if (shard_out) *shard_out = plan.shard(rank);
	if (n_rays_out) *n_rays_out = plan.n_rays(rank);
	return SPHIP_OK;
}

int sphip_device_count(const sphip_t* c) { return c ? (c->kids.empty() ? 1 : (int)c->kids.size()) : 0; }

int sphip_create_multi(const int* device_ids, int n_devices, sphip_t** out) {
	if (!out) return fail(nullptr, SPHIP_E_INVALID, "out is NULL");
	*out = nullptr;
	std::vector<int> ids;
	if (device_ids) {
		if (n_devices < 1 || n_devices > 64) return fail(nullptr, SPHIP_E_INVALID, "n_devices %d out of range [1,64]", n_devices);
		ids.assign(device_ids, device_ids + n_devices);
	} else if (const char* env = getenv("SPATH_HIP_DEVICES")) {
		for (const char* p = env; *p;) {
			char* end = nullptr;
			const long v = strtol(p, &end, 10);
			if (end == p) return fail(nullptr, SPHIP_E_INVALID, "SPATH_HIP_DEVICES=\"%s\" is not a comma-separated list of device numbers", env);
			ids.push_back((int)v);
			p = (*end == ',') ? end + 1 : end;
			if (*end && *end != ',') return fail(nullptr, SPHIP_E_INVALID, "SPATH_HIP_DEVICES=\"%s\" is not a comma-separated list of device numbers", env);
		}
		if (ids.empty() || ids.size() > 64) return fail(nullptr, SPHIP_E_INVALID, "SPATH_HIP_DEVICES lists %zu devices", ids.size());
	} else {
		int n = 0;
		const hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n <= 0)
			return fail(nullptr, SPHIP_E_DEVICE, "no HIP device available (%s)", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
		for (int i = 0; i < n; ++i) ids.push_back(i);
	}
	const char* want = getenv("SPATH_HIP_GATHER");
	// one device: the plain context, no exchange step (unless the RCCL exchange is asked for explicitly: a communicator of one)
	if (ids.size() == 1 && !(want && !strcmp(want, "rccl"))) return sphip_create(ids[0], out);
	sphip_ctx* c = new (std::nothrow) sphip_ctx();
	if (!c) return fail(nullptr, SPHIP_E_DEVICE, "out of host memory");
	bool distinct = true;
	for (size_t i = 0; i < ids.size(); ++i) for (size_t j = 0; j < i; ++j) distinct &= ids[i] != ids[j];
	for (int id : ids) {
		sphip_ctx* k = nullptr;
		if (sphip_create(id, &k) != SPHIP_OK) { sphip_destroy(c); return SPHIP_E_DEVICE; }     // g_create_error already says why
		c->kids.push_back(k);
	}
	sphip_ctx* root = c->kids[0];
	c->device = root->device;
	hipError_t e = hipSetDevice(root->device);
	if (e == hipSuccess) e = hipEventCreate(&c->ev_g0);
	if (e == hipSuccess) e = hipEventCreate(&c->ev_g1);
	for (size_t r = 0; r < c->kids.size() && e == hipSuccess; ++r) {
		hipEvent_t ev = nullptr;
		if ((e = hipSetDevice(c->kids[r]->device)) == hipSuccess && (e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) == hipSuccess) c->ev_tile.push_back(ev);
		// direct xGMI copies into the first device's gather buffer (not an error if already enabled or unsupported: the copy still works, staged)
		if (e == hipSuccess && c->kids[r]->device != root->device) (void)hipDeviceEnablePeerAccess(root->device, 0);
		(void)hipGetLastError();
	}
	if (e != hipSuccess) { fail(nullptr, SPHIP_E_DEVICE, "multi-device init failed: %s", hipGetErrorString(e)); sphip_destroy(c); return SPHIP_E_DEVICE; }
	// exchange: RCCL when every listed device is a different GPU and librccl loads; peer copies otherwise
	c->gather_kind = SPHIP_GATHER_PEER;
	if (distinct && !(want && !strcmp(want, "peer"))) {
		c->rccl_lib = load_rccl();
		if (c->rccl_lib) {
			c->comms.assign(ids.size(), nullptr);
			const int nrc = g_rccl.init_all(c->comms.data(), (int)ids.size(), ids.data());
			if (nrc == 0) c->gather_kind = SPHIP_GATHER_RCCL;
			else { c->comms.clear(); if (want && !strcmp(want, "rccl")) { fail(nullptr, SPHIP_E_DEVICE, "ncclCommInitAll failed: %s", g_rccl.errstr ? g_rccl.errstr(nrc) : "?"); sphip_destroy(c); return SPHIP_E_DEVICE; } }
		} else if (want && !strcmp(want, "rccl")) { fail(nullptr, SPHIP_E_DEVICE, "SPATH_HIP_GATHER=rccl but librccl could not be loaded: %s", dlerror()); sphip_destroy(c); return SPHIP_E_DEVICE; }
	} else if (want && !strcmp(want, "rccl") && !distinct) {
		fail(nullptr, SPHIP_E_INVALID, "SPATH_HIP_GATHER=rccl needs distinct devices (a communicator holds a GPU once)"); sphip_destroy(c); return SPHIP_E_INVALID;
	}
	char d[320];
	snprintf(d, sizeof d, "HIP - Path Tracing (%zu devices, first: %s; pixel-row tiles, %s gather)", ids.size(),
	         root->desc.c_str() + (root->desc.rfind("(") == std::string::npos ? 0 : root->desc.rfind("(") + 1), c->gather_kind == SPHIP_GATHER_RCCL ? "RCCL" : "peer-copy");
	c->desc = d;
	*out = c;
	return SPHIP_OK;
}

int sphip_create(int device_id, sphip_t** out) {
	if (!out) return fail(nullptr, SPHIP_E_INVALID, "out is NULL");
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return fail(nullptr, SPHIP_E_DEVICE, "no HIP device available (%s)", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
	if (device_id < 0 || device_id >= n) return fail(nullptr, SPHIP_E_INVALID, "device %d out of range [0,%d)", device_id, n);
	sphip_ctx* c = new (std::nothrow) sphip_ctx();
	if (!c) return fail(nullptr, SPHIP_E_DEVICE, "out of host memory");
	c->device = device_id;
	hipDeviceProp_t prop;
	if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
	    (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
		fail(nullptr, SPHIP_E_DEVICE, "device %d init failed: %s", device_id, hipGetErrorString(e));
		delete c;
		return SPHIP_E_DEVICE;
	}
	hipEvent_t* evs[6] = { &c->ev_k0, &c->ev_k1, &c->ev_u0, &c->ev_u1, &c->ev_d0, &c->ev_d1 };
	for (auto ev : evs) {
		if ((e = hipEventCreate(ev)) != hipSuccess) {
			fail(nullptr, SPHIP_E_DEVICE, "hipEventCreate failed: %s", hipGetErrorString(e));
			sphip_destroy(c);
			return SPHIP_E_DEVICE;
		}
	}
	char d[256];
	snprintf(d, sizeof d, "HIP - Path Tracing (%s, %s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
	c->desc = d;
	*out = c;
	return SPHIP_OK;
}

void sphip_destroy(sphip_t* c) {
	if (!c) return;
	if (!c->kids.empty()) {
		for (void* comm : c->comms) if (comm) (void)g_rccl.destroy(comm);
		(void)hipSetDevice(c->kids[0]->device);
		(void)hipDeviceSynchronize();
		DevBuf* mb[4] = { &c->gath, &c->gath_acc, &c->img, &c->img_acc };
		for (auto b : mb) if (b->p) (void)hipFree(b->p);
		if (c->ev_g0) (void)hipEventDestroy(c->ev_g0);
		if (c->ev_g1) (void)hipEventDestroy(c->ev_g1);
		for (size_t r = 0; r < c->ev_tile.size(); ++r) { (void)hipSetDevice(c->kids[r]->device); (void)hipEventDestroy(c->ev_tile[r]); }
		for (sphip_ctx* k : c->kids) sphip_destroy(k);
		delete c;
		return;
	}
	(void)hipSetDevice(c->device);
	if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
	DevBuf* bufs[] = { &c->tris, &c->mats, &c->scan, &c->filt, &c->bounds, &c->samp, &c->rays, &c->rgba, &c->accum, &c->counter, &c->work,
	                   &c->bvh_nodes, &c->bvh_rec, &c->bvh_idx, &c->sort_kv, &c->sort_hist, &c->bvh_meta, &c->cyl_rec, &c->cyl_cnt, &c->cyl_hdr, &c->prim, &c->cylm_rec, &c->cylm_hdr, &c->cylm_big };
	for (auto b : bufs) if (b->p) (void)hipFree(b->p);
	hipEvent_t evs[6] = { c->ev_k0, c->ev_k1, c->ev_u0, c->ev_u1, c->ev_d0, c->ev_d1 };
	for (auto ev : evs) if (ev) (void)hipEventDestroy(ev);
	if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
	delete c;
}

const char* sphip_last_error(const sphip_t* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

const char* sphip_description(const sphip_t* c) { return c ? c->desc.c_str() : "HIP - Path Tracing"; }

int sphip_set_scene(sphip_t* c, const float* tris, const float* mats, size_t n_tris) {
	if (!c) return SPHIP_E_INVALID;
	if (!tris || !mats || n_tris == 0 || n_tris > 0x7fffffffull) return fail(c, SPHIP_E_INVALID, "bad scene arguments (n_tris=%zu)", n_tris);
	if (!c->kids.empty()) return multi_set_scene(c, tris, mats, n_tris);
	HIP_TRY(c, hipSetDevice(c->device));
	int rc;
	if ((rc = ensure(c, c->tris, n_tris * 48)) || (rc = ensure(c, c->mats, n_tris * 24))) return rc;
	c->n_tris = n_tris;
	HIP_TRY(c, hipMemcpyAsync(c->tris.p, tris, n_tris * 48, hipMemcpyHostToDevice, c->own_stream));
	HIP_TRY(c, hipMemcpyAsync(c->mats.p, mats, n_tris * 24, hipMemcpyHostToDevice, c->own_stream));
	if ((rc = repack(c, c->own_stream))) return rc;
	HIP_TRY(c, hipStreamSynchronize(c->own_stream));   // tris/mats are borrowed: do not outlive the call
	return SPHIP_OK;
}

int sphip_set_scene_device(sphip_t* c, const void* d_tris, const void* d_mats, size_t n_tris, void* stream) {
	if (!c) return SPHIP_E_INVALID;
	if (!c->kids.empty()) return fail(c, SPHIP_E_STATE, "device-pointer entry points need a single-device context (sphip_create)");
	if (!d_tris || !d_mats || n_tris == 0 || n_tris > 0x7fffffffull) return fail(c, SPHIP_E_INVALID, "bad scene arguments (n_tris=%zu)", n_tris);
	HIP_TRY(c, hipSetDevice(c->device));
	hipStream_t st = (hipStream_t)stream;
	int rc;
	if ((rc = ensure(c, c->tris, n_tris * 48)) || (rc = ensure(c, c->mats, n_tris * 24))) return rc;
	c->n_tris = n_tris;
	HIP_TRY(c, hipMemcpyAsync(c->tris.p, d_tris, n_tris * 48, hipMemcpyDeviceToDevice, st));
	HIP_TRY(c, hipMemcpyAsync(c->mats.p, d_mats, n_tris * 24, hipMemcpyDeviceToDevice, st));
	return repack(c, st);
}

int sphip_render_device(sphip_t* c, const void* d_rays, size_t n_rays, const sphip_shard* shard, size_t image_width,
                        size_t n_samples, uint64_t seed, int mode, int flags, void* d_out_rgba, void* d_out_accum, void* stream) {
	if (!c) return SPHIP_E_INVALID;
	if (!c->kids.empty()) return fail(c, SPHIP_E_STATE, "device-pointer entry points need a single-device context (sphip_create)");
	HIP_TRY(c, hipSetDevice(c->device));
	c->timed_upload = c->timed_download = false;
	return launch_render(c, d_rays, n_rays, shard, image_width, n_samples, seed, mode, flags, d_out_rgba, d_out_accum, (hipStream_t)stream);
}

int sphip_viewport_device(sphip_t* c, const sphip_camera* cam, void* d_rays_out, void* stream) {
	if (!c) return SPHIP_E_INVALID;
	if (!c->kids.empty()) return fail(c, SPHIP_E_STATE, "device-pointer entry points need a single-device context (sphip_create)");
	HIP_TRY(c, hipSetDevice(c->device));
	return launch_viewport(c, cam, d_rays_out, (hipStream_t)stream);
}

int sphip_render_camera(sphip_t* c, const sphip_camera* cam, size_t n_samples, uint64_t seed, int mode, int flags,
                        uint8_t* out_rgba, float* out_accum) {
	if (!c) return SPHIP_E_INVALID;
	if (!cam || !out_rgba) return fail(c, SPHIP_E_INVALID, "null camera or output pointer");
	if (!c->kids.empty()) return multi_render(c, nullptr, cam, cam->res_x, cam->res_y, n_samples, seed, mode, flags, out_rgba, out_accum);
	HIP_TRY(c, hipSetDevice(c->device));
	const size_t n = (size_t)cam->res_x * cam->res_y;
	hipStream_t st = c->own_stream;
	int rc;
	if ((rc = ensure(c, c->rays, (n ? n : 1) * 24)) || (rc = ensure(c, c->rgba, (n ? n : 1) * 4))) return rc;
	if (out_accum && (rc = ensure(c, c->accum, n * 12))) return rc;
	if ((rc = launch_viewport(c, cam, c->rays.p, st))) return rc;
	if ((rc = launch_render(c, c->rays.p, n, nullptr, cam->res_x, n_samples, seed, mode, flags, c->rgba.p, out_accum ? c->accum.p : nullptr, st))) return rc;
	HIP_TRY(c, hipEventRecord(c->ev_d0, st));
	HIP_TRY(c, hipMemcpyAsync(out_rgba, c->rgba.p, n * 4, hipMemcpyDeviceToHost, st));
	if (out_accum) HIP_TRY(c, hipMemcpyAsync(out_accum, c->accum.p, n * 12, hipMemcpyDeviceToHost, st));
	HIP_TRY(c, hipEventRecord(c->ev_d1, st));
	HIP_TRY(c, hipStreamSynchronize(st));
	c->timed_upload = false;
	c->timed_download = true;
	return SPHIP_OK;
}

int sphip_closest_hit_device(sphip_t* c, const void* d_rays, size_t n_rays, const void* d_src_idx, int flags,
                             void* d_out_idx, void* d_out_dist, void* stream) {
	if (!c) return SPHIP_E_INVALID;
	if (!c->kids.empty()) return fail(c, SPHIP_E_STATE, "device-pointer entry points need a single-device context (sphip_create)");
	HIP_TRY(c, hipSetDevice(c->device));
	c->timed_upload = c->timed_download = false;
	return launch_render(c, d_rays, n_rays, nullptr, 0, 1, 0, kModeHits, flags, d_out_idx, d_out_dist, (hipStream_t)stream, (const int*)d_src_idx);
}

int sphip_render(sphip_t* c, const float* rays, size_t w, size_t h, size_t n_samples, uint64_t seed, int mode, int flags,
                 uint8_t* out_rgba, float* out_accum) {
	if (!c) return SPHIP_E_INVALID;
	if (!rays || !out_rgba || w == 0 || h == 0) return fail(c, SPHIP_E_INVALID, "bad render arguments (w=%zu h=%zu)", w, h);
	if (!c->kids.empty()) return multi_render(c, rays, nullptr, w, h, n_samples, seed, mode, flags, out_rgba, out_accum);
	HIP_TRY(c, hipSetDevice(c->device));
	const size_t n = w * h;
	hipStream_t st = c->own_stream;
	int rc;
	if ((rc = ensure(c, c->rays, n * 24)) || (rc = ensure(c, c->rgba, n * 4))) return rc;
	if (out_accum && (rc = ensure(c, c->accum, n * 12))) return rc;
	HIP_TRY(c, hipEventRecord(c->ev_u0, st));
	HIP_TRY(c, hipMemcpyAsync(c->rays.p, rays, n * 24, hipMemcpyHostToDevice, st));
	HIP_TRY(c, hipEventRecord(c->ev_u1, st));
	if ((rc = launch_render(c, c->rays.p, n, nullptr, w, n_samples, seed, mode, flags, c->rgba.p, out_accum ? c->accum.p : nullptr, st))) return rc;
	HIP_TRY(c, hipEventRecord(c->ev_d0, st));
	HIP_TRY(c, hipMemcpyAsync(out_rgba, c->rgba.p, n * 4, hipMemcpyDeviceToHost, st));
	if (out_accum) HIP_TRY(c, hipMemcpyAsync(out_accum, c->accum.p, n * 12, hipMemcpyDeviceToHost, st));
	HIP_TRY(c, hipEventRecord(c->ev_d1, st));
	HIP_TRY(c, hipStreamSynchronize(st));           // blocking, like every reference backend (main.cpp:70-83)
	c->timed_upload = c->timed_download = true;
	return SPHIP_OK;
}

int sphip_selftest_device(sphip_t* c, int what, const void* in, size_t n, void* out) {
	if (!c) return SPHIP_E_INVALID;
	static const size_t in_b[7] = { 4, 4, 20, 40, 60, 12, 48 }, out_b[7] = { 8, 4, 16, 12, 4, 4, 8 };
	if (what < 0 || what > 6 || !in || !out || n == 0 || n > 0x7fffffffull) return fail(c, SPHIP_E_INVALID, "bad selftest arguments (what=%d n=%zu)", what, n);
	sphip_ctx* k = c->kids.empty() ? c : c->kids[0];
	HIP_TRY(c, hipSetDevice(k->device));
	void *d_in = nullptr, *d_out = nullptr;
	HIP_TRY(c, hipMalloc(&d_in, n * in_b[what]));
	hipError_t e = hipMalloc(&d_out, n * out_b[what]);
	if (e == hipSuccess) e = hipMemcpy(d_in, in, n * in_b[what], hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		if (what == 6) hipLaunchKernelGGL(sp::cylm256::k_selftest_cylm, dim3((unsigned)n), dim3(64), 0, k->own_stream, (const float*)d_in, (uint32_t)n, (float*)d_out);
		else hipLaunchKernelGGL(sp::k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, k->own_stream, what, (const void*)d_in, (uint32_t)n, d_out);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(k->own_stream);
	if (e == hipSuccess) e = hipMemcpy(out, d_out, n * out_b[what], hipMemcpyDeviceToHost);
	(void)hipFree(d_in);
	if (d_out) (void)hipFree(d_out);
	if (e != hipSuccess) return fail(c, SPHIP_E_DEVICE, "selftest failed: %s", hipGetErrorString(e));
	return SPHIP_OK;
}

int sphip_selftest_stage1(sphip_t* c, const float* rays, size_t n_rays, uint32_t* out_words, uint32_t* out_tri, int32_t* out_order, uint32_t* tiles_out) {
	if (!c) return SPHIP_E_INVALID;
	if (!c->kids.empty()) return fail(c, SPHIP_E_STATE, "sphip_selftest_stage1 needs a single-device context (sphip_create)");
	if (!c->have_scene) return fail(c, SPHIP_E_STATE, "sphip_selftest_stage1 called before a scene was set");
	if (!tiles_out) return fail(c, SPHIP_E_INVALID, "tiles_out is NULL");
	HIP_TRY(c, hipSetDevice(c->device));
	hipStream_t st = c->own_stream;
	int rc = ensure_cylm(c, st);
	if (rc) return rc;
	uint32_t hdr[8];
	HIP_TRY(c, hipMemcpyAsync(hdr, c->cylm_hdr.p, sizeof hdr, hipMemcpyDeviceToHost, st));
	HIP_TRY(c, hipStreamSynchronize(st));
	const uint32_t tiles = hdr[6];
	const uint32_t T = c->cylm_wide ? sp::cylm512::kMTile : sp::cylm256::kMTile, W = c->cylm_wide ? sp::cylm512::kMWords : sp::cylm256::kMWords;
	const uint32_t grp8 = (c->cylm_wide ? sp::cylm512::kMGrp : sp::cylm256::kMGrp) == 8u ? 1u : 0u;
	*tiles_out = tiles | (T << 20) | (grp8 << 31);         // tiles of the stream (low 20 bits), triangles per tile, bit 31: one bit per octet (else per quad)
	if (!out_words) return SPHIP_OK;
	if (!rays || !out_order || n_rays == 0 || n_rays % 64 || n_rays > 0x7fffffffull) return fail(c, SPHIP_E_INVALID, "bad stage-1 selftest arguments (n_rays=%zu)", n_rays);
	const size_t words_b = n_rays * tiles * 2 * W * sizeof(uint32_t), tri_b = n_rays * tiles * 2 * (T / 64) * sizeof(uint32_t),
	             order_b = (size_t)tiles * T * sizeof(int32_t);
	void *d_rays = nullptr, *d_words = nullptr, *d_order = nullptr, *d_tri = nullptr;
	hipError_t e = hipMalloc(&d_rays, n_rays * 24);
	if (e == hipSuccess) e = hipMalloc(&d_words, words_b);
	if (e == hipSuccess && out_tri) e = hipMalloc(&d_tri, tri_b);
	if (e == hipSuccess) e = hipMalloc(&d_order, order_b);
	if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, n_rays * 24, hipMemcpyHostToDevice, st);
	if (e == hipSuccess) {
		sp::CylStream cs{ (const float4*)c->cylm_rec.p, (const uint32_t*)c->cylm_hdr.p, (const float4*)c->cylm_big.p };
		if (c->cylm_wide)
			hipLaunchKernelGGL(sp::cylm512::k_selftest_stage1, dim3((unsigned)(n_rays / 64)), dim3(64), 0, st, (const float*)d_rays, (uint32_t)n_rays, cs,
		                   (const unsigned int*)c->bounds.p, (uint32_t*)d_words, (uint32_t*)d_tri, (int*)d_order);
		else
			hipLaunchKernelGGL(sp::cylm256::k_selftest_stage1, dim3((unsigned)(n_rays / 64)), dim3(64), 0, st, (const float*)d_rays, (uint32_t)n_rays, cs,
		                   (const unsigned int*)c->bounds.p, (uint32_t*)d_words, (uint32_t*)d_tri, (int*)d_order);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(out_words, d_words, words_b, hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipMemcpyAsync(out_order, d_order, order_b, hipMemcpyDeviceToHost, st);
	if (e == hipSuccess && out_tri) e = hipMemcpyAsync(out_tri, d_tri, tri_b, hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (d_rays) (void)hipFree(d_rays);
	if (d_words) (void)hipFree(d_words);
	if (d_order) (void)hipFree(d_order);
	if (d_tri) (void)hipFree(d_tri);
	if (e != hipSuccess) return fail(c, SPHIP_E_DEVICE, "stage-1 selftest failed: %s", hipGetErrorString(e));
	return SPHIP_OK;
}

int sphip_get_stats(sphip_t* c, sphip_stats* out) {
	if (!c || !out) return SPHIP_E_INVALID;
	if (!c->kids.empty()) return multi_get_stats(c, out);
	if (!c->have_render) return fail(c, SPHIP_E_STATE, "no render has been issued yet");
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipEventSynchronize(c->ev_k1));
	float ms = 0.0f;
	HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_k0, c->ev_k1));
	c->stats.kernel_ms = ms;
	c->stats.upload_ms = c->stats.download_ms = 0.0;
	if (c->timed_upload) { HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_u0, c->ev_u1)); c->stats.upload_ms = ms; }
	if (c->timed_download) { HIP_TRY(c, hipEventSynchronize(c->ev_d1)); HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_d0, c->ev_d1)); c->stats.download_ms = ms; }
	unsigned long long scans = 0;
	HIP_TRY(c, hipMemcpyAsync(&scans, c->counter.p, sizeof scans, hipMemcpyDeviceToHost, c->last_stream));
	HIP_TRY(c, hipStreamSynchronize(c->last_stream));
	c->stats.scans_executed = scans;
#ifdef SP_BVH_STATS
	{
		unsigned long long x[3] = {0, 0, 0};
		(void)hipMemcpy(x, c->counter.p, sizeof x, hipMemcpyDeviceToHost);
		fprintf(stderr, "[bvh stats] scans=%llu steps/scan=%.1f leaves/scan=%.1f\n", x[0], (double)x[1] / (double)(x[0] ? x[0] : 1), (double)x[2] / (double)(x[0] ? x[0] : 1));
	}
#endif
#ifdef SP_PHASE_TIMERS
	{
		unsigned long long x[16] = {};
		(void)hipMemcpy(x, c->counter.p, sizeof x, hipMemcpyDeviceToHost);
		const double tot = (double)(x[8] + x[9] + x[10] + x[11] + x[12]);
		fprintf(stderr, "[phase timers] wave-scans %llu; wave lifetime inside the scan by phase: stage 1 %.1f %%, list building + DMA issue %.1f %%, re-test rounds %.1f %%, exact turns %.1f %%, tile barrier %.1f %% (%.0f cycles per wave-scan)\n",
		        x[13], 100.0 * x[8] / tot, 100.0 * x[9] / tot, 100.0 * x[10] / tot, 100.0 * x[11] / tot, 100.0 * x[12] / tot, tot / (double)(x[13] ? x[13] : 1));
	}
#endif
#ifdef SP_FILTER_STATS
	{
		unsigned long long x[6] = {0, 0, 0, 0, 0, 0};
		(void)hipMemcpy(x, c->counter.p, sizeof x, hipMemcpyDeviceToHost);
		if (c->stats.kernel_variant >= 9)
			fprintf(stderr, "[cyl stats] group bits set=%llu stage-2 rounds(per wave)=%llu wave-tiles=%llu exact tests=%llu -> bits/lane/tile=%.3f rounds/tile=%.2f exact per bit=%.3f lane utilisation in stage 2=%.3f wave-wide exact turns=%llu (%.2f per round, %.3f of their lanes used)\n",
			        x[1], x[2], x[3], x[4], (double)x[1] / (64.0 * (double)x[3]), (double)x[2] / (double)x[3], (double)x[4] / (double)x[1], (double)x[4] / (64.0 * (double)x[2]),
			        x[5], (double)x[5] / (double)x[2], (double)x[4] / (64.0 * (double)x[5]));
		else
		fprintf(stderr, "[filter stats] survivors=%llu rounds(sum of per-wave max)=%llu wave_flushes=%llu overflows=%llu -> survivors/lane/flush=%.3f rounds/flush=%.2f\n",
		        x[1], x[2], x[3], x[4], (double)x[1] / (64.0 * (double)x[3]), (double)x[2] / (double)x[3]);
	}
#endif
	c->stats.n_devices = 1; c->stats.gather_kind = SPHIP_GATHER_NONE; c->stats.gather_ms = 0.0; c->stats.kernel_ms_min = c->stats.kernel_ms;
	*out = c->stats;
	return SPHIP_OK;
}

} // extern "C"
