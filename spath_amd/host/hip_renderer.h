// hip_renderer -- the MI355X backend as a peer of cpu_renderer / cl_renderer / vk_renderer.
// Same shape as the reference's per-backend headers (src/cpu_renderer.h:23-25): one factory.
#pragma once

#include "spath_iface.h"

namespace hip_renderer {
	// Returns a new renderer owned by the caller (the reference wraps it in std::unique_ptr,
	// src/main.cpp:242-244).  Throws std::runtime_error when no usable HIP device exists, like the
	// reference's GPU peers do from their constructors (src/cl_renderer.cpp:155-187).
	// One GPU: device 0 (or the devices listed in the environment variable SPATH_HIP_DEVICES, e.g. "0,1,2,3").
	extern scene::renderer* get(const int w, const int h);
	// Opt-in: several GPUs of the node behind one renderer -- pixel-row tiles dealt round-robin, one gather to the first device,
	// same image bit for bit as on one GPU.  An explicit device list (a device may be listed more than once), or every visible GPU.
	extern scene::renderer* get_on(const int w, const int h, const int* device_ids, const int n_devices);
	extern scene::renderer* get_all_devices(const int w, const int h);
	extern int device_count(scene::renderer* r);

	// Optional knobs of this backend (not part of the reference interface): the RNG seed of the next
	// frames and the C-ABI flags word (kernel variant, primary-hit reuse; see include/spath_hip.h).
	extern void set_seed(scene::renderer* r, unsigned long long seed);
	extern void set_flags(scene::renderer* r, int flags);
	// get_viewport + render (or render_flat) with the viewport generated on the device from the renderer's own
	// camera (bit-identical rays, no 24 B/pixel upload).  Returns false if r is not a hip renderer.
	extern bool render_own_viewport(scene::renderer* r, const geom::triangle* tris, const scene::material* mats, const size_t n_tris,
	                                const size_t n_samples, scene::bitmap& out, const bool flat);
	// kernel milliseconds and closest-hit scans of the last frame (0 if r is not a hip renderer)
	extern bool last_stats(scene::renderer* r, double* kernel_ms, unsigned long long* scans);
}
