"""BASELINE.json configs[2] at FULL size (10k triangles, 1920x1080, 256 spp): the exact-only scan and the two-stage
filter scan must produce the same frame bit for bit -- 2.65e13 ray-triangle pairs through both."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene, view
ctx = capi.Context(0)
t, m = scene.closed_room(10000)
w, h, spp = 1920, 1080, 256
rays = view.Camera(w, h).get_viewport()
dev = torch.device("cuda")
d_t, d_m, d_r = torch.from_numpy(t).to(dev), torch.from_numpy(m).to(dev), torch.from_numpy(rays).to(dev)
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), 10000, 0)
res = {}
for name, var in (("auto (rpl_cylm)", 0), ("rpl_cylw4s (f32 stage 1)", 15), ("rpl_cyl4s (per-lane stage 2)", 13), ("rpl_filter2s", 6), ("rpl_lds (exact only)", 2)):
    img = torch.zeros(w * h, 4, dtype=torch.uint8, device=dev); acc = torch.zeros(w * h, 3, dtype=torch.float32, device=dev)
    ctx.render_device(d_r.data_ptr(), w * h, spp, img.data_ptr(), seed=1, flags=var, d_out_accum=acc.data_ptr())
    torch.cuda.synchronize(); st = ctx.stats()
    res[name] = (img.cpu().numpy(), acc.cpu().numpy())
    print(f"{name}: {st['kernel_ms']/1e3:.2f} s, scans {st['scans_executed']}, rgba sha256 {hashlib.sha256(res[name][0].tobytes()).hexdigest()[:16]}, "
          f"accum sha256 {hashlib.sha256(res[name][1].tobytes()).hexdigest()[:16]}", flush=True)
b = res["rpl_lds (exact only)"]
for name, a in res.items():
    if a is not b:
        print(f"{name} vs exact only:  RGBA8 identical: {np.array_equal(a[0], b[0])}  float accumulators identical: {np.array_equal(a[1], b[1])}"
              f"  max |diff|: {float(np.abs(a[1] - b[1]).max())}", flush=True)
