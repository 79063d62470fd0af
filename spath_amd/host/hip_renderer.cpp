// The C++ adapter between the reference's plugin interface and the C ABI of libspath_hip.so.
//
// hip_r derives from basic_renderer exactly like the reference's pt_r / cl_r / vk_r do
// (src/cpu_renderer.cpp:186-202, src/cl_renderer.cpp:90-257), so the five camera virtuals come from
// the shared mix-in and only get_description / render_flat / render are implemented here.
// Compiles against the reference's own headers (-DSPATH_REFERENCE_HEADERS -I<spath>/src) or against
// the compatible declarations in spath_iface.h.
#include "hip_renderer.h"

#include "spath_hip.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace {

static_assert(sizeof(geom::triangle) == 48 && sizeof(geom::ray) == 24 && sizeof(scene::material) == 24 && sizeof(scene::RGBA) == 4,
              "the C ABI takes the reference's packed structs as plain float arrays");

struct hip_r : public basic_renderer {
	sphip_t* ctx;
	std::string desc;
	unsigned long long seed;
	int flags;
	// scene cache: the reference's GPU peer re-uploads every frame (src/cl_renderer.cpp:210-214); tris/mats are
	// borrowed pointers that may be reused with new contents, so the key is a content hash, not the address
	unsigned long long scene_hash;
	size_t scene_n;
	sphip_stats stats;
	bool have_stats;

	// ids == 0: every visible GPU of the node (or the list in SPATH_HIP_DEVICES) behind this one renderer object: the frame is
	// dealt to them as interleaved pixel-row tiles and reassembled on the first (include/spath_hip.h: sphip_create_multi)
	hip_r(const int x, const int y, const int* ids, const int n_ids) : basic_renderer(x, y), ctx(0), seed(1), flags(0), scene_hash(0), scene_n(0), have_stats(false) {
		if (sphip_create_multi(ids, n_ids, &ctx) != SPHIP_OK)
			throw std::runtime_error(std::string("hip_renderer: ") + sphip_last_error(0));
		desc = sphip_description(ctx);
		std::memset(&stats, 0, sizeof stats);
	}

	virtual ~hip_r() { sphip_destroy(ctx); }

	virtual const char* get_description(void) const { return desc.c_str(); }

	void check(int rc, const char* what) {
		if (rc != SPHIP_OK) throw std::runtime_error(std::string("hip_renderer: ") + what + ": " + sphip_last_error(ctx));
	}

	static unsigned long long fnv(const void* p, size_t n, unsigned long long h) {
		const unsigned char* b = (const unsigned char*)p;
		for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
		return h;
	}

	void upload_scene(const geom::triangle* tris, const scene::material* mats, const size_t n_tris) {
		unsigned long long h = fnv(tris, n_tris * sizeof(geom::triangle), 1469598103934665603ull);
		h = fnv(mats, n_tris * sizeof(scene::material), h);
		if (h != scene_hash || n_tris != scene_n) {
			check(sphip_set_scene(ctx, (const float*)tris, (const float*)mats, n_tris), "set_scene");
			scene_hash = h; scene_n = n_tris;
		}
	}

	// the camera as the C ABI wants it; the trig values are recomputed with the same float std::cos/std::sin calls the
	// reference's camera makes (src/view.h:77-80,87-92) -- its cached copies are private
	sphip_camera camera_args() const {
		sphip_camera c;
		c.pos[0] = vc.pos.x; c.pos[1] = vc.pos.y; c.pos[2] = vc.pos.z;
		c.cos_y = std::cos(vc.angle.y); c.sin_y = std::sin(vc.angle.y);
		c.cos_x = std::cos(vc.angle.x); c.sin_x = std::sin(vc.angle.x);
		c.focal = vc.focal;
		c.res_x = (uint32_t)vc.res_x; c.res_y = (uint32_t)vc.res_y;
		return c;
	}

	void frame_own_viewport(const geom::triangle* tris, const scene::material* mats, const size_t n_tris, const size_t n_samples,
	                        scene::bitmap& out, const int mode) {
		upload_scene(tris, mats, n_tris);
		out.res_x = vc.res_x;
		out.res_y = vc.res_y;
		out.values.resize(out.res_x * out.res_y);
		const sphip_camera c = camera_args();
		check(sphip_render_camera(ctx, &c, n_samples, seed, mode, flags, (uint8_t*)out.values.data(), 0), "render_camera");
		have_stats = sphip_get_stats(ctx, &stats) == SPHIP_OK;
	}

	void frame(const view::viewport& vp, const geom::triangle* tris, const scene::material* mats, const size_t n_tris,
	           const size_t n_samples, scene::bitmap& out, const int mode) {
		upload_scene(tris, mats, n_tris);
		// first ensure that the bitmap is of correct size (src/cpu_renderer.cpp:120-122)
		out.res_x = vp.res_x;
		out.res_y = vp.res_y;
		out.values.resize(out.res_x * out.res_y);
		if (vp.rays.size() != out.values.size()) throw std::runtime_error("hip_renderer: viewport size and ray count disagree");
		check(sphip_render(ctx, (const float*)vp.rays.data(), vp.res_x, vp.res_y, n_samples, seed, mode, flags,
		                   (uint8_t*)out.values.data(), 0), "render");
		have_stats = sphip_get_stats(ctx, &stats) == SPHIP_OK;
	}

	virtual void render_flat(const view::viewport& vp, const geom::triangle* tris, const scene::material* mats, const size_t n_tris, const size_t n_samples, scene::bitmap& out) {
		frame(vp, tris, mats, n_tris, n_samples ? n_samples : 1, out, SPHIP_MODE_FLAT);     // n_samples unused (src/cpu_renderer.cpp:81)
	}

	virtual void render(const view::viewport& vp, const geom::triangle* tris, const scene::material* mats, const size_t n_tris, const size_t n_samples, scene::bitmap& out) {
		frame(vp, tris, mats, n_tris, n_samples, out, SPHIP_MODE_PT);
	}
};

} // namespace

namespace hip_renderer {
	// One GPU: device 0, or the devices listed in SPATH_HIP_DEVICES.  Several GPUs behind one renderer are opt-in (get_on /
	// get_all_devices): the cross-device exchange has not run on a multi-GPU node yet.
	scene::renderer* get(const int w, const int h) {
		if (std::getenv("SPATH_HIP_DEVICES")) return new hip_r(w, h, 0, 0);
		const int first = 0;
		return new hip_r(w, h, &first, 1);
	}

	scene::renderer* get_all_devices(const int w, const int h) {
		return new hip_r(w, h, 0, 0);
	}

	scene::renderer* get_on(const int w, const int h, const int* device_ids, const int n_devices) {
		return new hip_r(w, h, device_ids, n_devices);
	}

	int device_count(scene::renderer* r) {
		hip_r* p = dynamic_cast<hip_r*>(r);
		return p ? sphip_device_count(p->ctx) : 0;
	}

	void set_seed(scene::renderer* r, unsigned long long seed) {
		if (hip_r* p = dynamic_cast<hip_r*>(r)) p->seed = seed;
	}

	void set_flags(scene::renderer* r, int flags) {
		if (hip_r* p = dynamic_cast<hip_r*>(r)) p->flags = flags;
	}

	bool render_own_viewport(scene::renderer* r, const geom::triangle* tris, const scene::material* mats, const size_t n_tris,
	                         const size_t n_samples, scene::bitmap& out, const bool flat) {
		hip_r* p = dynamic_cast<hip_r*>(r);
		if (!p) return false;
		p->frame_own_viewport(tris, mats, n_tris, n_samples ? n_samples : 1, out, flat ? SPHIP_MODE_FLAT : SPHIP_MODE_PT);
		return true;
	}

	bool last_stats(scene::renderer* r, double* kernel_ms, unsigned long long* scans) {
		hip_r* p = dynamic_cast<hip_r*>(r);
		if (!p || !p->have_stats) return false;
		if (kernel_ms) *kernel_ms = p->stats.kernel_ms;
		if (scans) *scans = p->stats.scans_executed;
		return true;
	}
}
