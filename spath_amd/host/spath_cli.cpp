// Headless front end for the renderer plugin interface: what the reference's GLUT shell does per frame
// (src/main.cpp:70-83: get_viewport -> render / render_flat -> show the bitmap), minus the window.
// Keys of the interactive shell become flags: camera moves (w/a/s/d, mouse -> --mov/--rot/--focal),
// '+'/'-' -> --spp, 'p' -> --mode.  The image goes to a binary PPM or a raw RGBA file.
//
//   spath_cli [--scene default|FILE.bin] [--w 640 --h 480] [--spp 128] [--mode pt|flat]
//             [--mov x y z] [--rot x y z] [--focal f] [--seed n] [--flags n] [--primary-reuse] [--out image.ppm|image.rgba] [--frames n]
//             [--device-viewport] [--gpus n | --devices 0,1,... | --all-gpus]
// --out: .ppm (binary P6), .png (8-bit RGB, stored deflate blocks: no compression library needed), anything else = raw RGBA8
#include "hip_renderer.h"
#include "spath_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

namespace {

geom::vec3 sub(const geom::vec3& a, const geom::vec3& b) { return geom::vec3(a.x - b.x, a.y - b.y, a.z - b.z); }

void set_flat_normal(geom::triangle& t) {   // unit((v1-v0) x (v2-v0)), the caller's job in the reference too (main.cpp:214-215)
	const geom::vec3 a = sub(t.v1, t.v0), b = sub(t.v2, t.v0);
	const geom::vec3 c(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
	const real l = std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z);
	t.n = geom::vec3(c.x / l, c.y / l, c.z / l);
}

// the demo scene of the reference (values: SURVEY.md Appendix C): pyramid face, floor, area light, back wall
void default_scene(std::vector<geom::triangle>& t, std::vector<scene::material>& m) {
	static const float v[7][9] = {
		{ 0, 0, 1, 0.5f, -0.5f, 0, -0.5f, -0.5f, 0 },
		{ 20, -1, 20, -20, -1, -20, -20, -1, 20 }, { 20, -1, 20, 20, -1, -20, -20, -1, -20 },
		{ 0.75f, 0.75f, 0.75f, -0.75f, 0.75f, 0.75f, 0.75f, 0.75f, -0.75f }, { -0.75f, 0.75f, 0.75f, -0.75f, 0.75f, -0.75f, 0.75f, 0.75f, -0.75f },
		{ 1.25f, 0.5f, 1, 1.25f, -1, 1, -1.25f, -1, 1 }, { 1.25f, 0.5f, 1, -1.25f, -1, 1, -1.25f, 0.5f, 1 } };
	static const float refl[7][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 1 } };
	t.resize(7); m.resize(7);
	for (int i = 0; i < 7; ++i) {
		t[i].v0 = geom::vec3(v[i][0], v[i][1], v[i][2]); t[i].v1 = geom::vec3(v[i][3], v[i][4], v[i][5]); t[i].v2 = geom::vec3(v[i][6], v[i][7], v[i][8]);
		set_flat_normal(t[i]);
		m[i].reflectance_color = geom::vec3(refl[i][0], refl[i][1], refl[i][2]);
		const float e = (i == 3 || i == 4) ? 1.0f : 0.0f;
		m[i].emittance_color = geom::vec3(e, e, e);
	}
}

// ---- minimal PNG writer: 8-bit RGB, filter 0 on every row, zlib stream of STORED deflate blocks (RFC 1950/1951/2083)
uint32_t crc32_of(const unsigned char* p, size_t n, uint32_t crc = 0) {
	static uint32_t tab[256];
	static bool init = false;
	if (!init) { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; tab[i] = c; } init = true; }
	crc = ~crc;
	for (size_t i = 0; i < n; ++i) crc = tab[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
	return ~crc;
}

void png_chunk(FILE* o, const char* type, const std::vector<unsigned char>& data) {
	unsigned char len[4] = { (unsigned char)(data.size() >> 24), (unsigned char)(data.size() >> 16), (unsigned char)(data.size() >> 8), (unsigned char)data.size() };
	std::fwrite(len, 1, 4, o);
	std::vector<unsigned char> td(type, type + 4);
	td.insert(td.end(), data.begin(), data.end());
	std::fwrite(td.data(), 1, td.size(), o);
	const uint32_t c = crc32_of(td.data(), td.size());
	unsigned char cb[4] = { (unsigned char)(c >> 24), (unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c };
	std::fwrite(cb, 1, 4, o);
}

void write_png(FILE* o, const scene::bitmap& bmp) {
	const size_t w = bmp.res_x, h = bmp.res_y;
	std::vector<unsigned char> raw;                       // filter byte + RGB per row (row 0 = top, as stored)
	raw.reserve(h * (1 + 3 * w));
	for (size_t j = 0; j < h; ++j) {
		raw.push_back(0);
		for (size_t i = 0; i < w; ++i) { const scene::RGBA& p = bmp.values[j * w + i]; raw.push_back(p.r); raw.push_back(p.g); raw.push_back(p.b); }
	}
	std::vector<unsigned char> z;
	z.push_back(0x78); z.push_back(0x01);                 // zlib header, no compression
	uint32_t a = 1, b = 0;                                // Adler-32 of the raw data
	for (size_t i = 0; i < raw.size(); ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
	for (size_t pos = 0; pos < raw.size() || pos == 0; ) {
		const size_t n = std::min<size_t>(65535, raw.size() - pos);
		const bool last = pos + n >= raw.size();
		z.push_back(last ? 1 : 0);
		z.push_back((unsigned char)(n & 0xff)); z.push_back((unsigned char)(n >> 8));
		z.push_back((unsigned char)(~n & 0xff)); z.push_back((unsigned char)((~n >> 8) & 0xff));
		z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
		pos += n;
		if (last) break;
	}
	const uint32_t ad = (b << 16) | a;
	z.push_back((unsigned char)(ad >> 24)); z.push_back((unsigned char)(ad >> 16)); z.push_back((unsigned char)(ad >> 8)); z.push_back((unsigned char)ad);
	static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
	std::fwrite(sig, 1, 8, o);
	std::vector<unsigned char> hd = { (unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w,
	                                  (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h, 8, 2, 0, 0, 0 };
	png_chunk(o, "IHDR", hd);
	png_chunk(o, "IDAT", z);
	png_chunk(o, "IEND", std::vector<unsigned char>());
}

bool load_scene(const char* path, std::vector<geom::triangle>& t, std::vector<scene::material>& m) {   // 'SPSC' file of spath_amd/scene.py
	FILE* f = std::fopen(path, "rb");
	if (!f) return false;
	uint32_t hdr[2];
	bool ok = std::fread(hdr, 4, 2, f) == 2 && hdr[0] == 0x43535053u;
	if (ok) {
		t.resize(hdr[1]); m.resize(hdr[1]);
		ok = std::fread(t.data(), sizeof(geom::triangle), hdr[1], f) == hdr[1] && std::fread(m.data(), sizeof(scene::material), hdr[1], f) == hdr[1];
	}
	std::fclose(f);
	return ok;
}

} // namespace

int main(int argc, char** argv) {
	try {
		std::string scene_arg = "default", mode = "pt", out_path;
		bool device_viewport = false;
		std::vector<int> devices;                                    // empty: one GPU (hip_renderer::get: device 0 or SPATH_HIP_DEVICES)
		bool all_gpus = false;
		int w = 640, h = 480, frames = 1, flags = 0;                 // window default of the reference (main.cpp:238-239)
		size_t spp = 128;                                            // main.cpp:44
		unsigned long long seed = 1;
		std::vector<std::pair<char, geom::vec3> > moves;
		for (int i = 1; i < argc; ++i) {
			const std::string k = argv[i];
			auto need = [&](int n) { if (i + n >= argc) throw std::runtime_error("missing value after " + k); };
			if (k == "--scene") { need(1); scene_arg = argv[++i]; }
			else if (k == "--w") { need(1); w = std::atoi(argv[++i]); }
			else if (k == "--h") { need(1); h = std::atoi(argv[++i]); }
			else if (k == "--spp") { need(1); spp = (size_t)std::atoll(argv[++i]); }
			else if (k == "--mode") { need(1); mode = argv[++i]; }
			else if (k == "--seed") { need(1); seed = std::strtoull(argv[++i], 0, 0); }
			else if (k == "--flags") { need(1); flags = std::atoi(argv[++i]); }
			else if (k == "--primary-reuse") flags |= SPHIP_FLAG_PRIMARY_REUSE;   // one primary scan per pixel (identical image)
			else if (k == "--frames") { need(1); frames = std::atoi(argv[++i]); }
			else if (k == "--out") { need(1); out_path = argv[++i]; }
			else if (k == "--device-viewport") device_viewport = true;
			else if (k == "--all-gpus") all_gpus = true;
			else if (k == "--gpus") { need(1); const int n = std::atoi(argv[++i]); devices.clear(); for (int d = 0; d < n; ++d) devices.push_back(d); }
			else if (k == "--devices") { need(1); devices.clear(); for (const char* p = argv[++i]; *p;) { char* e = 0; devices.push_back((int)std::strtol(p, &e, 10)); if (e == p) throw std::runtime_error("bad --devices list"); p = *e ? e + 1 : e; } }
			else if (k == "--mov" || k == "--rot") { need(3); moves.push_back(std::make_pair(k[2], geom::vec3(std::atof(argv[i + 1]), std::atof(argv[i + 2]), std::atof(argv[i + 3])))); i += 3; }
			else if (k == "--focal") { need(1); moves.push_back(std::make_pair('f', geom::vec3(std::atof(argv[++i]), 0, 0))); }
			else throw std::runtime_error("unknown argument " + k);
		}
		std::vector<geom::triangle> tris;
		std::vector<scene::material> mats;
		if (scene_arg == "default") default_scene(tris, mats);
		else if (!load_scene(scene_arg.c_str(), tris, mats)) throw std::runtime_error("cannot read scene file " + scene_arg);

		std::unique_ptr<scene::renderer> r(!devices.empty() ? hip_renderer::get_on(w, h, devices.data(), (int)devices.size())      // main.cpp:242-244
		                                   : all_gpus ? hip_renderer::get_all_devices(w, h) : hip_renderer::get(w, h));
		hip_renderer::set_seed(r.get(), seed);
		hip_renderer::set_flags(r.get(), flags);
		std::printf("Current renderer: %s [%d device(s)]\n", r->get_description(), hip_renderer::device_count(r.get()));     // main.cpp:30-32
		for (size_t k = 0; k < moves.size(); ++k) {
			if (moves[k].first == 'm') r->set_delta_mov(moves[k].second);
			else if (moves[k].first == 'r') r->set_delta_rot(moves[k].second);
			else r->set_delta_focal(moves[k].second.x);
		}
		view::viewport vp;
		scene::bitmap bmp;
		for (int f = 0; f < frames; ++f) {
			const auto t0 = std::chrono::steady_clock::now();
			if (device_viewport) {                                       // rays generated on the GPU, never uploaded
				hip_renderer::render_own_viewport(r.get(), tris.data(), mats.data(), tris.size(), spp, bmp, mode != "pt");
			} else {
				r->get_viewport(vp);                                     // main.cpp:74
				if (mode == "pt") r->render(vp, tris.data(), mats.data(), tris.size(), spp, bmp);
				else r->render_flat(vp, tris.data(), mats.data(), tris.size(), spp, bmp);
			}
			const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			double kms = 0; unsigned long long scans = 0;
			hip_renderer::last_stats(r.get(), &kms, &scans);
			std::printf("Done (%.3fs) frame %d: %dx%d, %zu spp, %zu triangles, kernel %.3f ms, %llu scans, %.1f Mray/s nominal\n", s, f, w, h,
			            spp, tris.size(), kms, scans, mode == "pt" ? (double)w * h * spp * 5 / s / 1e6 : (double)w * h / s / 1e6);
		}
		if (!out_path.empty()) {
			FILE* o = std::fopen(out_path.c_str(), "wb");
			if (!o) throw std::runtime_error("cannot write " + out_path);
			if (out_path.size() > 4 && out_path.substr(out_path.size() - 4) == ".ppm") {
				std::fprintf(o, "P6\n%zu %zu\n255\n", bmp.res_x, bmp.res_y);
				for (size_t i = 0; i < bmp.values.size(); ++i) std::fwrite(&bmp.values[i], 1, 3, o);   // row 0 = top, as stored
			} else if (out_path.size() > 4 && out_path.substr(out_path.size() - 4) == ".png") {
				write_png(o, bmp);
			} else {
				std::fwrite(bmp.values.data(), sizeof(scene::RGBA), bmp.values.size(), o);
			}
			std::fclose(o);
		}
	} catch (const std::exception& e) {
		std::fprintf(stderr, "Exception: %s\n", e.what());               // main.cpp:263-264
		return 1;
	}
	return 0;
}
