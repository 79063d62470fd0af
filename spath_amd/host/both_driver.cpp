// TEST INFRASTRUCTURE.  Registers the reference's cpu_renderer and this repo's hip_renderer in one
// std::vector<scene::renderer*> exactly like the reference's main() does (src/main.cpp:242-248) and
// drives both through the same interface calls.  Compiled only against the reference's own headers
// (spath_amd/host/Makefile refcheck); the binary lands in oracle/_ref/ (git-ignored).
//   spath_both scene.bin W H SPP out_prefix   -> <prefix>.<k>.flat.rgba and <prefix>.<k>.pt.rgba per renderer k
#include "cpu_renderer.h"
#include "hip_renderer.h"

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

unsigned int std::thread::hardware_concurrency() noexcept {
	const char* e = std::getenv("ORACLE_THREADS");
	const int v = e ? std::atoi(e) : 0;
	return v > 0 ? (unsigned)v : 8u;
}

int main(int argc, char** argv) {
	if (argc < 6) { std::fprintf(stderr, "usage: %s scene.bin W H SPP out_prefix\n", argv[0]); return 1; }
	try {
		const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
		const size_t spp = (size_t)std::atoll(argv[4]);
		std::vector<geom::triangle> t;
		std::vector<scene::material> m;
		FILE* f = std::fopen(argv[1], "rb");
		uint32_t hdr[2];
		if (!f || std::fread(hdr, 4, 2, f) != 2 || hdr[0] != 0x43535053u) throw std::runtime_error("bad scene file");
		t.resize(hdr[1]); m.resize(hdr[1]);
		if (std::fread(t.data(), sizeof(geom::triangle), hdr[1], f) != hdr[1] || std::fread(m.data(), sizeof(scene::material), hdr[1], f) != hdr[1]) throw std::runtime_error("short scene file");
		std::fclose(f);
		std::unique_ptr<scene::renderer> pt_r(cpu_renderer::get(w, h)), hip_r(hip_renderer::get(w, h));
		std::vector<scene::renderer*> all_renderers;
		all_renderers.push_back(&(*pt_r));
		all_renderers.push_back(&(*hip_r));
		for (size_t k = 0; k < all_renderers.size(); ++k) {
			scene::renderer* r = all_renderers[k];
			std::printf("Current renderer: %s\n", r->get_description());
			r->set_delta_mov(geom::vec3(0.1, 0.05, -0.2));
			r->set_delta_rot(geom::vec3(0.0, 0.15, 0.0));
			view::viewport vp;
			scene::bitmap bmp;
			r->get_viewport(vp);
			const char* modes[2] = { "flat", "pt" };
			for (int mo = 0; mo < 2; ++mo) {
				if (mo == 0) r->render_flat(vp, t.data(), m.data(), t.size(), spp, bmp);
				else r->render(vp, t.data(), m.data(), t.size(), spp, bmp);
				const std::string p = std::string(argv[5]) + "." + std::to_string(k) + "." + modes[mo] + ".rgba";
				FILE* o = std::fopen(p.c_str(), "wb");
				if (!o) throw std::runtime_error("cannot write " + p);
				std::fwrite(bmp.values.data(), sizeof(scene::RGBA), bmp.values.size(), o);
				std::fclose(o);
			}
		}
	} catch (const std::exception& e) {
		std::fprintf(stderr, "Exception: %s\n", e.what());
		return 1;
	}
	return 0;
}
