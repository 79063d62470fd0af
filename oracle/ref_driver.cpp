// TEST INFRASTRUCTURE ONLY -- not part of the shipped product path.
//
// Headless driver around the UNMODIFIED reference CPU backend.  It is compiled
// (oracle/Makefile) together with /root/reference/src/cpu_renderer.cpp, taken
// from where it lies; no reference source is copied into this repository and
// the resulting binary goes to oracle/_ref/ (git-ignored).
//
// What it replaces: the interactive GLUT shell (reference src/main.cpp:55-181,
// 236-262) which cannot be built here (no GL/glut.h).  It drives the plugin
// interface exactly like gl::displayFunc does (main.cpp:70-83):
//   renderer::get_viewport(vp) -> renderer::render / render_flat(vp, tris, mats, n, spp, bmp)
//
// Thread pinning: cpu_renderer.cpp:124 sizes its thread pool with
// std::thread::hardware_concurrency() and seeds one LCG per thread (:147), so
// the image depends on the core count.  The definition below pre-empts the
// libstdc++ one at link time and reads ORACLE_THREADS, which pins T without
// touching the reference.
//
// usage:
//   spath_ref render   scene.bin W H SPP out.rgba [cam...]
//   spath_ref flat     scene.bin W H SPP out.rgba [cam...]
//   spath_ref viewport -         W H 0   out.rays [cam...]
//   spath_ref kat      -         0 0 0   out.txt
// cam... = optional "mov x y z", "rot x y z", "focal f" groups applied in order
// through set_delta_mov / set_delta_rot / set_delta_focal (basic_renderer.h:37-49).
// optional "rays file" replaces the viewport with P*6 raw floats from a file.
//
// scene.bin: u32 magic 'SPSC' (0x43535053), u32 n_tris, n*12 f32 (v0 v1 v2 n), n*6 f32 (refl emit)

#include "cpu_renderer.h"
#include "frand.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

unsigned int std::thread::hardware_concurrency() noexcept {
	const char* e = std::getenv("ORACLE_THREADS");
	const int v = e ? std::atoi(e) : 0;
	return v > 0 ? (unsigned)v : 8u;
}

namespace {

static_assert(sizeof(geom::triangle) == 48, "triangle layout");
static_assert(sizeof(geom::ray) == 24, "ray layout");
static_assert(sizeof(scene::material) == 24, "material layout");
static_assert(sizeof(scene::RGBA) == 4, "pixel layout");

bool read_scene(const char* path, std::vector<geom::triangle>& t, std::vector<scene::material>& m) {
	FILE* f = std::fopen(path, "rb");
	if (!f) return false;
	uint32_t hdr[2];
	if (std::fread(hdr, 4, 2, f) != 2 || hdr[0] != 0x43535053u) { std::fclose(f); return false; }
	t.resize(hdr[1]);
	m.resize(hdr[1]);
	bool ok = std::fread(t.data(), sizeof(geom::triangle), hdr[1], f) == hdr[1]
	       && std::fread(m.data(), sizeof(scene::material), hdr[1], f) == hdr[1];
	std::fclose(f);
	return ok;
}

uint32_t bits(float x) { uint32_t u; std::memcpy(&u, &x, 4); return u; }

void p3(FILE* o, const char* tag, const geom::vec3& v) {
	std::fprintf(o, "%s %08x %08x %08x\n", tag, bits(v.x), bits(v.y), bits(v.z));
}

// header-level known answers straight from the reference's inline functions
int dump_kat(const char* out_path) {
	FILE* o = std::fopen(out_path, "w");
	if (!o) return 2;
	// frand.h:53-63
	for (uint32_t s : {0u, 1u, 7u, 12345u}) {
		frand::seed_dist d(s);
		std::fprintf(o, "seed_dist %u", s);
		for (int i = 0; i < 8; ++i) {
			const double v = d();
			uint64_t u; std::memcpy(&u, &v, 8);
			std::fprintf(o, " %016llx", (unsigned long long)u);
		}
		std::fprintf(o, "\n");
	}
	// geom.h:164-177
	{
		const geom::vec3 ns[3] = { geom::vec3(0, 1, 0), geom::vec3(0, 0, -1), geom::vec3(0.6f, -0.48f, 0.64f) };
		for (int k = 0; k < 3; ++k) {
			frand::seed_dist d(7 + k);
			for (int i = 0; i < 6; ++i) {
				const geom::vec3 v = geom::rand_unit_vec(ns[k], d);
				std::fprintf(o, "rand_unit_vec %d %d %08x %08x %08x\n", k, i, bits(v.x), bits(v.y), bits(v.z));
			}
		}
	}
	// std::cos / std::sin (float overloads) on every angle the 15-bit LCG can produce
	// (geom.h:168-171); dumped as two running FNV-1a-64 hashes plus spot values
	{
		uint64_t hs = 1469598103934665603ull, hc = hs;
		for (int k = 0; k <= 32767; ++k) {
			const double r = 1.0 * k / 32767.0;
			const real a = 1.0 * r * geom::PI * 2.0, b = 1.0 * r * geom::PI * 0.5;
			const float v[4] = { std::sin(a), std::sin(b), std::cos(a), std::cos(b) };
			for (int j = 0; j < 2; ++j) { uint32_t u = bits(v[j]); for (int q = 0; q < 4; ++q) { hs ^= (u >> (8 * q)) & 0xff; hs *= 1099511628211ull; } }
			for (int j = 2; j < 4; ++j) { uint32_t u = bits(v[j]); for (int q = 0; q < 4; ++q) { hc ^= (u >> (8 * q)) & 0xff; hc *= 1099511628211ull; } }
			if (k % 4096 == 5)
				std::fprintf(o, "trig %d %08x %08x %08x %08x %08x %08x\n", k, bits(a), bits(b), bits(v[0]), bits(v[1]), bits(v[2]), bits(v[3]));
		}
		std::fprintf(o, "trig_hash %016llx %016llx\n", (unsigned long long)hs, (unsigned long long)hc);
	}
	// geom.h:192-195, 197-222
	{
		geom::triangle t;
		t.v0 = geom::vec3(0.0, 0.0, 1.0); t.v1 = geom::vec3(0.5, -0.5, 0.0); t.v2 = geom::vec3(-0.5, -0.5, 0.0);
		geom::flat_normal(t);
		p3(o, "flat_normal", t.n);
		const float ys[4] = { -0.25f, 0.5f, -0.5f, -0.1f };
		for (int i = 0; i < 4; ++i) {
			geom::ray r; r.pos = geom::vec3(0.0, ys[i], -3.0); r.dir = geom::vec3(0.0, 0.0, 1.0);
			geom::vec3 pt;
			const real d = geom::ray_intersect(r, t, pt);
			std::fprintf(o, "ray_intersect %d %08x", i, bits(d));
			if (d > 0) std::fprintf(o, " %08x %08x %08x", bits(pt.x), bits(pt.y), bits(pt.z));
			std::fprintf(o, "\n");
		}
	}
	// scene.h:32-39
	{
		const float vs[10] = { -0.1f, 0.0f, 0.001f, 0.00196f, 0.00197f, 0.5f, 0.998f, 0.9981f, 1.0f, 7.0f };
		std::fprintf(o, "vec3_RGBA");
		for (int i = 0; i < 10; ++i) {
			const scene::RGBA c = scene::vec3_RGBA(geom::vec3(vs[i], vs[i], vs[i]));
			std::fprintf(o, " %u", (unsigned)c.r);
		}
		std::fprintf(o, "\n");
	}
	// constants (cpu_renderer.cpp:27,60,63,67; geom.h:160,198)
	{
		const real p = 1.0 / (geom::PI * 2.0), ip = 1.0 / geom::PI, invp = 1.0 / p, eps = 0.00000000000001, mx = 1000000000000.0;
		const double ie = 1.0 / eps;
		uint64_t u; std::memcpy(&u, &ie, 8);
		std::fprintf(o, "consts %08x %08x %08x %08x %08x %016llx\n", bits(p), bits(ip), bits(invp), bits(eps), bits(mx), (unsigned long long)u);
	}
	std::fclose(o);
	return 0;
}

} // namespace

int main(int argc, char** argv) {
	if (argc < 7) {
		std::fprintf(stderr, "usage: %s render|flat|viewport|kat scene.bin W H SPP out [mov x y z] [rot x y z] [focal f] [rays file]\n", argv[0]);
		return 1;
	}
	const std::string mode = argv[1];
	const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
	const size_t spp = (size_t)std::atoll(argv[5]);
	const char* out_path = argv[6];
	if (mode == "kat") return dump_kat(out_path);

	std::unique_ptr<scene::renderer> r(cpu_renderer::get(w, h));   // main.cpp:242
	const char* rays_path = 0;
	for (int i = 7; i < argc; ) {
		const std::string k = argv[i];
		if (k == "mov" && i + 3 < argc) { r->set_delta_mov(geom::vec3(std::atof(argv[i+1]), std::atof(argv[i+2]), std::atof(argv[i+3]))); i += 4; }
		else if (k == "rot" && i + 3 < argc) { r->set_delta_rot(geom::vec3(std::atof(argv[i+1]), std::atof(argv[i+2]), std::atof(argv[i+3]))); i += 4; }
		else if (k == "focal" && i + 1 < argc) { r->set_delta_focal((real)std::atof(argv[i+1])); i += 2; }
		else if (k == "rays" && i + 1 < argc) { rays_path = argv[i+1]; i += 2; }
		else { std::fprintf(stderr, "bad camera argument '%s'\n", argv[i]); return 1; }
	}
	view::viewport vp;
	r->get_viewport(vp);                                           // main.cpp:74
	if (rays_path) {
		FILE* f = std::fopen(rays_path, "rb");
		if (!f || std::fread(vp.rays.data(), sizeof(geom::ray), vp.rays.size(), f) != vp.rays.size()) { std::fprintf(stderr, "cannot read rays\n"); return 2; }
		std::fclose(f);
	}
	FILE* o = std::fopen(out_path, "wb");
	if (!o) { std::fprintf(stderr, "cannot open %s\n", out_path); return 2; }
	if (mode == "viewport") {
		std::fwrite(vp.rays.data(), sizeof(geom::ray), vp.rays.size(), o);
		std::fclose(o);
		return 0;
	}
	std::vector<geom::triangle> tris;
	std::vector<scene::material> mats;
	if (!read_scene(argv[2], tris, mats)) { std::fprintf(stderr, "cannot read scene %s\n", argv[2]); return 2; }
	scene::bitmap bmp;
	const auto t0 = std::chrono::steady_clock::now();
	if (mode == "render") r->render(vp, tris.data(), mats.data(), tris.size(), spp, bmp);          // main.cpp:76
	else if (mode == "flat") r->render_flat(vp, tris.data(), mats.data(), tris.size(), spp, bmp);  // main.cpp:78
	else { std::fprintf(stderr, "unknown mode\n"); return 1; }
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	std::fwrite(bmp.values.data(), sizeof(scene::RGBA), bmp.values.size(), o);
	std::fclose(o);
	std::fprintf(stderr, "{\"mode\": \"%s\", \"w\": %d, \"h\": %d, \"spp\": %zu, \"n_tris\": %zu, \"threads\": %u, \"seconds\": %.6f}\n",
		mode.c_str(), w, h, spp, tris.size(), std::thread::hardware_concurrency(), secs);
	return 0;
}
