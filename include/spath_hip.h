/* spath_hip.h -- C ABI of the MI355X (gfx950) path-tracing backend for Emanem/spath.
 *
 * This is the drop-in boundary: plain C types, caller-allocated outputs, no exceptions, no C++
 * or torch types.  It is what a `hip_renderer` peer of the reference's cpu_renderer binds to
 * (see spath_amd/host/hip_renderer.cpp and INTEGRATION.md).  Each entry point names the reference
 * interface it stands behind (paths relative to the reference repository root).
 *
 * Data layouts are the reference's tightly packed float structs:
 *   ray       6 x f32  pos.xyz dir.xyz                  geom::ray        src/geom.h:179-182
 *   triangle 12 x f32  v0 v1 v2 n                       geom::triangle   src/geom.h:185-190
 *   material  6 x f32  reflectance.rgb emittance.rgb    scene::material  src/scene.h:47-50
 *   pixel     4 x u8   r g b a(=0)                      scene::RGBA      src/scene.h:25-30
 * Pixel index = i + j*W, row 0 = top (src/view.h:112).
 *
 * Every function returning int returns 0 on success and a negative SPHIP_E_* code otherwise;
 * sphip_last_error() then describes the failure.  The C++ adapter turns a non-zero status into
 * std::runtime_error, which is how the reference's GPU peers report failure
 * (src/cl_renderer.cpp:155-187, src/vk_renderer.cpp:318-349).
 */
#ifndef SPATH_HIP_H
#define SPATH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPHIP_ABI_VERSION 3

typedef struct sphip_ctx sphip_t;

enum {
	SPHIP_OK = 0,
	SPHIP_E_INVALID = -1,   /* bad argument (NULL pointer, n_samples == 0, ...) */
	SPHIP_E_DEVICE = -2,    /* HIP runtime error (no device, allocation failure, launch failure) */
	SPHIP_E_STATE = -3      /* call order error (render before set_scene) */
};

/* render modes: which renderer virtual is being served */
enum {
	SPHIP_MODE_FLAT = 0,    /* renderer::render_flat  src/renderer.h:31, body src/cpu_renderer.cpp:81-101 */
	SPHIP_MODE_PT = 1       /* renderer::render       src/renderer.h:32, body src/cpu_renderer.cpp:29-79,118-184 */
};

/* flags (bitwise or) */
enum {
	SPHIP_KERNEL_AUTO = 0,          /* let the library pick the scan kernel */
	SPHIP_KERNEL_MASK = 0xff,       /* low byte: explicit kernel variant (see sphip_kernel_name) for A/B runs;
	                                   every variant produces bit-identical images */
	SPHIP_FLAG_PRIMARY_REUSE = 0x100,/* scan the (identical) primary ray of a pixel once for all its samples
	                                   (src/cpu_renderer.cpp:74-76 re-scans it); identical image, ~1/5 fewer scans.
	                                   With the two-stage kernels: a closest-hit pre-pass, one scan per pixel, then the
	                                   path-tracing launch (stats: one launch more).  OFF by default: the default executes
	                                   every scan the reference executes */
	SPHIP_FLAG_CHUNKS_SHIFT = 16,   /* bits 16..23: number of sample chunks of a path-traced launch, 0 = let the library
	                                   choose.  A frame (or shard) with too few pixels to fill the GPU several times over is
	                                   launched as (pixel, sample chunk) lanes; each sample's radiance goes to a scratch buffer
	                                   and a second kernel adds them up per pixel in sample order (src/cpu_renderer.cpp:74-76),
	                                   so the image is bit-identical whatever the number of chunks.  1 = never split. */
	SPHIP_FLAG_CHUNKS_MASK = 0xff0000,
	SPHIP_FLAG_ACCEL = 0x200         /* OPT-IN acceleration structure (linear BVH).  Changes the work
	                                   definition: the reference tests every triangle (README.md:23).  Same strict triangle
	                                   test and tie rule, so every geometric hit is reproduced bit for bit; what it cannot
	                                   reproduce are the reference's rounding-noise accepts on rays almost coplanar with a
	                                   far-away triangle.  Never used by bench.py's headline figure. */
};

/* Pixel-shard descriptor: which global pixel the k-th ray of a shard is.
 *   global_pixel(k) = pixel_base + (k / tile_px) * tile_stride_px + (k % tile_px)
 * A whole image is {0, npix, 0}.  Row tiles dealt round-robin to G GPUs (rank r, tile of R rows of
 * width W): {r*R*W, R*W, G*R*W}.  The global index keys the counter RNG, so an image does not
 * depend on how it was sharded. */
typedef struct {
	uint64_t pixel_base;
	uint64_t tile_px;
	uint64_t tile_stride_px;
} sphip_shard;

typedef struct {
	double   kernel_ms;        /* device time of the last render's kernels (hipEvent, on the launch stream) */
	double   upload_ms;        /* host-pointer path only: H2D of rays (+ scene when it changed) */
	double   download_ms;      /* host-pointer path only: D2H of the image */
	uint64_t scans_executed;   /* closest-hit scans (one ray against all triangles) the last render ran */
	uint64_t n_tris;
	uint64_t n_pixels;
	uint32_t kernel_variant;   /* variant that actually ran */
	uint32_t n_launches;
	/* multi-device contexts (sphip_create_multi); a single-device context reports 1, 0, 0, kernel_ms */
	uint32_t n_devices;        /* devices the last render ran on */
	uint32_t gather_kind;      /* SPHIP_GATHER_* used to bring the row tiles to the first device */
	double   gather_ms;        /* first tile copy enqueued -> image assembled on the first device (hipEvent on its stream) */
	double   kernel_ms_min;    /* kernel_ms is the slowest device's, this the fastest's */
} sphip_stats;

enum {
	SPHIP_GATHER_NONE = 0,     /* one device */
	SPHIP_GATHER_RCCL = 1,     /* ncclSend/ncclRecv group over xGMI, single-process communicator (ncclCommInitAll) */
	SPHIP_GATHER_PEER = 2      /* hipMemcpyPeerAsync per device (also what runs when a device id is listed twice, or RCCL is absent) */
};

/* ---- lifetime.  Replaces X_renderer::get(w,h) construction (src/cpu_renderer.cpp:205-209) for the
 * device-owning part; device buffers are cached grow-only in the context like
 * src/cl_renderer.cpp:107-112 and released by sphip_destroy. */
int  sphip_create(int device_id, sphip_t** out);
/* All the GPUs of one node behind ONE context, so that a renderer object registered beside cpu_renderer
 * (src/main.cpp:242-248, interface src/renderer.h:24-36) drives every device: the framebuffer is dealt to the devices as
 * interleaved pixel-row tiles (round-robin, sphip_plan_*), every device holds the whole scene and renders its tiles on its
 * own stream from its own host thread, the RGBA8 tiles are gathered to the first device (RCCL over xGMI; peer copies as the
 * fallback), un-permuted there and read back once.  The image is bit-identical to a one-device render (pixel-keyed RNG).
 *   device_ids == NULL: every visible device, or the comma-separated list in the environment variable SPATH_HIP_DEVICES.
 *   A device may be listed more than once (several shards on one GPU: how a one-GPU box exercises this path).
 *   SPATH_HIP_GATHER=rccl|peer overrides the choice of exchange.
 * The host-pointer entry points (sphip_set_scene, sphip_render, sphip_render_camera, sphip_get_stats, sphip_description,
 * sphip_destroy) accept such a context; the device-pointer entry points need a single-device context (SPHIP_E_STATE). */
int  sphip_create_multi(const int* device_ids, int n_devices, sphip_t** out);
int  sphip_device_count(const sphip_t* ctx);        /* devices behind ctx (1 for sphip_create) */
void sphip_destroy(sphip_t* ctx);
const char* sphip_last_error(const sphip_t* ctx);   /* ctx may be NULL: error of a failed sphip_create */
const char* sphip_description(const sphip_t* ctx);  /* renderer::get_description  src/renderer.h:26 */
int  sphip_abi_version(void);
const char* sphip_kernel_name(int variant);         /* NULL when the variant does not exist */
const char* sphip_build_info(void);                 /* "src=<16 hex digits>": hash of the sources and compiler flags this library was built
                                                       from (__graft_entry__.source_hash), "src=unknown" for a hand-made build; profiler-derived
                                                       figures under profiles/ carry the same stamp */
int  sphip_kernel_available(int variant);           /* 1 when this build of the library carries the variant (the shipped build: the
                                                       default scan, the exact-only scans, one f32 filter scan for A/B runs and the
                                                       opt-in BVH; -DSP_ALL_VARIANTS builds: every generation), else 0 */

/* ---- host-pointer path: exactly what renderer::render / render_flat receive
 * (src/renderer.h:31-32): borrowed host arrays, valid only during the call; blocking. */
int sphip_set_scene(sphip_t* ctx, const float* tris, const float* mats, size_t n_tris);
int sphip_render(sphip_t* ctx, const float* rays, size_t w, size_t h, size_t n_samples,
                 uint64_t seed, int mode, int flags,
                 uint8_t* out_rgba /* w*h*4 */, float* out_accum /* w*h*3 or NULL */);

/* ---- device-resident path: pointers are HIP device pointers on ctx's device, `stream` is a
 * hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing; outputs are complete
 * when the stream reaches the end of the enqueued work.  Used for HBM-resident benchmarking and for
 * pixel-row-tile sharding across GPUs (one context per GPU).  Stream order is the only ordering: issue
 * sphip_set_scene_device and the renders that use that scene on the same stream (or synchronise in between);
 * d_tris / d_mats are copied, so they may be freed once the stream has passed the call. */
int sphip_set_scene_device(sphip_t* ctx, const void* d_tris, const void* d_mats, size_t n_tris, void* stream);
int sphip_render_device(sphip_t* ctx, const void* d_rays /* n_rays*6 f32: the shard's rays */, size_t n_rays,
                        const sphip_shard* shard /* NULL = {0, n_rays, 0} */, size_t image_width,
                        size_t n_samples, uint64_t seed, int mode, int flags,
                        void* d_out_rgba /* n_rays*4 u8 */, void* d_out_accum /* n_rays*3 f32 or NULL */,
                        void* stream);

/* ---- device-side viewport generation (camera -> rays without the 24 B/pixel upload).
 * Same arithmetic as view::camera::get_viewport (src/view.h:94-132), float operation for float operation, so the
 * rays are bit-identical to what renderer::get_viewport hands to render().  The camera's trigonometric values are
 * computed by the caller exactly like the reference does on the host (src/view.h:77-80,87-92: std::cos/std::sin of
 * angle.y and angle.x) and passed in. */
typedef struct {
	float pos[3];                 /* view::camera::pos   (src/view.h:70) */
	float cos_y, sin_y, cos_x, sin_x;
	float focal;                  /* view::camera::focal (src/view.h:72) */
	uint32_t res_x, res_y;
} sphip_camera;

int sphip_viewport_device(sphip_t* ctx, const sphip_camera* cam, void* d_rays_out /* res_x*res_y*6 f32 */, void* stream);

/* get_viewport + render / render_flat in one call, rays never leave the device: equals
 * sphip_render(ctx, rays_of(cam), cam->res_x, cam->res_y, ...).  Blocking; host output pointers. */
int sphip_render_camera(sphip_t* ctx, const sphip_camera* cam, size_t n_samples, uint64_t seed, int mode, int flags,
                        uint8_t* out_rgba, float* out_accum);

/* The closest-hit scan on its own (the loop of src/cpu_renderer.cpp:36-49 for each ray): for ray k writes the
 * index of the nearest accepted triangle (or -1) and its distance d (MAX_VALUE_DIST = 1e12f on a miss).
 * d_src_idx (may be NULL) gives each ray's idx_source, the triangle to skip (:40-41); NULL means -1 for all.
 * Used by the parity tests to compare scan kernels hit for hit; asynchronous like sphip_render_device. */
int sphip_closest_hit_device(sphip_t* ctx, const void* d_rays, size_t n_rays, const void* d_src_idx /* n_rays i32 or NULL */,
                             int flags, void* d_out_idx /* n_rays i32 */, void* d_out_dist /* n_rays f32 */, void* stream);

/* The row-tile plan, pure host arithmetic (no GPU needed): tiles of `tile_rows` image rows are dealt round-robin, tile t to
 * device t mod n_devices.  sphip_plan_tile_rows: the largest tile height <= 8 that gives every device the same number of
 * whole tiles (8 when none does).  sphip_plan_shard: device `rank`'s sphip_shard and ray count; returns SPHIP_E_INVALID on
 * nonsense arguments. */
int sphip_plan_tile_rows(size_t height, int n_devices);
int sphip_plan_shard(size_t width, size_t height, int n_devices, size_t tile_rows, int rank, sphip_shard* shard_out, size_t* n_rays_out);

/* TEST-ONLY: evaluates one device function of the path for n caller-supplied inputs (host pointers, blocking), so that the
 * device arithmetic can be compared with the oracle directly rather than only through whole renders.
 *   what 0  sincos_glibc   (std::sin/std::cos of geom.h:170-173)   in f32[n]                          out f32[2n] sin, cos
 *        1  recip_ieee     (1.0/a of geom.h:206)                   in f32[n]                          out f32[n]
 *        2  philox_uniforms (replaces frand.h:53-63)               in u32[5n] seed lo, hi, pixel, sample, depth   out f64[2n]
 *        3  rand_unit_vec  (geom.h:164-177, the two draws given)   in f64[5n] n.xyz, r1, r2           out f32[3n]
 *        4  ray_tri_strict (geom.h:197-222)                        in f32[15n] pos dir v0 v1 v2       out f32[n] distance or -1
 *        5  vec3_rgba      (scene.h:32-39)                         in f32[3n]                         out u32[n]
 *        6  the f16 matrix-pipe side product of sp_cylm_scan.h      in f32[12n] 5 triangle values, 5 ray values, P_a (a half), 0
 *                                                                  out f32[2n] the instruction's result, the same 16 products summed in double */
int sphip_selftest_device(sphip_t* ctx, int what, const void* in, size_t n, void* out);

/* TEST-ONLY: stage 1 of the default scan ALONE -- the conservative reject that decides which (ray, triangle) pairs ever reach
 * geom::ray_intersect (src/geom.h:197-222) -- for n_rays host rays (a multiple of 64) against the context's scene, formed
 * exactly as the render kernels form it (same ray setup, same fragment code, same tiles).  A test can then assert, pair by
 * pair, that no pair the reference accepts has its bit clear, instead of observing the filter only through the closest hit.
 *   *tiles_out         bits 0-19: tiles of the scene's stream, bits 20-30: T = triangles per tile, bit 31: one survivor bit per OCTET (two groups
 *                      of four) instead of per group of four (call with out_words = NULL first to size the outputs); W words per ray block
 *                      below: T / 256 with group bits, max(1, T / 512) with octet bits
 *   out_words[((k * tiles + t) * 2 + rb) * W + w]   word w of "lane" k (k = 64 b + l) for tile t.  Group bits: bit 31 - (4 f + j) set = the group of
 *                      four triangles 8 (8 w + f) + 2 j + (l >> 5) of tile t SURVIVES for ray 64 b + (l & 31) + 32 rb (f < 8, j < 4).  Octet bits:
 *                      bit 31 - (2 f + q) set = the groups 8 (16 w + f) + 4 q + (l >> 5) and that + 2 survive (f < 16, q < 2)
 *   out_tri[((k * tiles + t) * 2 + rb) * (T / 64) + f / 2]   (may be NULL) the same side products tested per TRIANGLE with the triangle's own
 *                      cylinder radius (the per-pair form of the test): bit 31 - (16 (f & 1) + 4 j + i) set = triangle
 *                      32 f + 8 j + 4 (l >> 5) + i of tile t survives for that ray (f < T / 32).  Stronger than the group / octet bit: a set
 *                      triangle bit implies the set bit of its group / octet
 *   out_order[t * T + 4 g + u]            index of the triangle at place u of group g of tile t (n_tris = padding; a triangle that
 *                      appears nowhere is in the "big" class: no filter, tested by every ray)
 * Blocking; host pointers; single-device contexts. */
int sphip_selftest_stage1(sphip_t* ctx, const float* rays, size_t n_rays, uint32_t* out_words, uint32_t* out_tri, int32_t* out_order, uint32_t* tiles_out);

/* Blocks until the last render on this context has finished, then reports its figures. */
int sphip_get_stats(sphip_t* ctx, sphip_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* SPATH_HIP_H */
