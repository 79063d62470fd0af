// Interface types the HIP backend plugs into.
//
// With -DSPATH_REFERENCE_HEADERS (and -I<spath>/src) this header simply pulls in the reference's own
// renderer.h / basic_renderer.h, so hip_renderer.cpp compiles straight into the reference tree as a
// fourth backend.  Without it -- the reference tree is not available on the GPU box --
// it declares binary-compatible stand-ins written for this repository: same namespaces, names,
// layouts (48/24/24/4-byte structs) and virtual signatures (reference src/renderer.h:24-36,
// src/basic_renderer.h:25-54, src/view.h:28-132, src/scene.h:25-50, src/geom.h:24-190), which is all
// the adapter and the headless CLI need.
#pragma once

#ifdef SPATH_REFERENCE_HEADERS
#include "basic_renderer.h"
#else

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

typedef float real;

namespace geom {
struct vec3 {
	real x, y, z;
	vec3() : x(0), y(0), z(0) {}
	vec3(real a, real b, real c) : x(a), y(b), z(c) {}
};
struct ray { vec3 pos, dir; };
struct triangle { vec3 v0, v1, v2, n; };
static_assert(sizeof(vec3) == 12 && sizeof(ray) == 24 && sizeof(triangle) == 48, "geom layouts");
} // namespace geom

namespace view {
struct viewport {
	size_t res_x, res_y;
	std::vector<geom::ray> rays;
};

// pinhole camera: image plane 1.0 high through the camera origin, +x on the left, row 0 on top,
// rotate about X then Y, translate last (behaviour of reference view.h:94-132, checked bit for bit
// against it by tests/test_host_adapter.py)
struct camera {
	geom::vec3 pos, angle;
	real focal;
	size_t res_x, res_y;

	camera(size_t w, size_t h, real f = 2.0f) : pos(0.0f, 0.0f, -3.0f), angle(), focal(f), res_x(w), res_y(h) { update_angles_trig_values(); }

	void update_angles_trig_values() { cy = std::cos(angle.y); sy = std::sin(angle.y); cx = std::cos(angle.x); sx = std::sin(angle.x); }

	geom::vec3 rel_move(const geom::vec3& v) const {
		const geom::vec3 a(v.x, v.y * cx + v.z * -sx, v.y * sx + v.z * cx);          // about X
		return geom::vec3(a.x * cy + a.z * sy, a.y, a.x * -sy + a.z * cy);          // then about Y
	}

	void get_viewport(viewport& out) const {
		out.res_x = res_x; out.res_y = res_y;
		out.rays.resize(res_x * res_y);
		const real xs = (real)(1.0 * res_x / res_y), ys = 1.0f;
		const real xmax = (real)(xs / 2.0), xstep = xs / res_x, hx = (real)(xstep / 2.0);
		const real ymax = (real)(ys / 2.0), ystep = ys / res_y, hy = (real)(ystep / 2.0);
		for (size_t j = 0; j < res_y; ++j)
			for (size_t i = 0; i < res_x; ++i) {
				geom::ray& r = out.rays[i + j * res_x];
				const geom::vec3 p(xmax - xstep * (int)i - hx, ymax - ystep * (int)j - hy, 0.0f);
				const geom::vec3 d(p.x + 0.0f, p.y + 0.0f, 0.0f + focal);
				const real l = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
				const geom::vec3 pr = rel_move(p), dr = rel_move(geom::vec3(d.x / l, d.y / l, d.z / l));
				r.pos = geom::vec3(pr.x + pos.x, pr.y + pos.y, pr.z + pos.z);
				r.dir = dr;
			}
	}

private:
	real cy, sy, cx, sx;
};
} // namespace view

namespace scene {
struct RGBA { uint8_t r, g, b, a; };
struct bitmap {
	size_t res_x, res_y;
	std::vector<RGBA> values;
};
struct material { geom::vec3 reflectance_color, emittance_color; };
static_assert(sizeof(RGBA) == 4 && sizeof(material) == 24, "scene layouts");

class renderer {
public:
	virtual const char* get_description(void) const = 0;
	virtual void set_viewport_size(const int w, const int h) = 0;
	virtual void set_delta_mov(const geom::vec3& m) = 0;
	virtual void set_delta_rot(const geom::vec3& r) = 0;
	virtual void set_delta_focal(const real f) = 0;
	virtual void get_viewport(view::viewport& vp) = 0;
	virtual void render_flat(const view::viewport& vp, const geom::triangle* tris, const scene::material* mats, const size_t n_tris, const size_t n_samples, scene::bitmap& out) = 0;
	virtual void render(const view::viewport& vp, const geom::triangle* tris, const scene::material* mats, const size_t n_tris, const size_t n_samples, scene::bitmap& out) = 0;
	virtual ~renderer() {}
};
} // namespace scene

// camera-owning partial implementation shared by every backend
class basic_renderer : public scene::renderer {
protected:
	view::camera vc;

public:
	basic_renderer(const int x, const int y) : vc(x, y) {}
	virtual void set_viewport_size(const int w, const int h) { vc.res_x = w; vc.res_y = h; }
	virtual void set_delta_mov(const geom::vec3& m) { const geom::vec3 d = vc.rel_move(m); vc.pos = geom::vec3(vc.pos.x + d.x, vc.pos.y + d.y, vc.pos.z + d.z); }
	virtual void set_delta_rot(const geom::vec3& r) { vc.angle = geom::vec3(vc.angle.x + r.x, vc.angle.y + r.y, vc.angle.z + r.z); vc.update_angles_trig_values(); }
	virtual void set_delta_focal(const real f) { vc.focal += f; }
	virtual void get_viewport(view::viewport& vp) { vc.get_viewport(vp); }
};

#endif // SPATH_REFERENCE_HEADERS
