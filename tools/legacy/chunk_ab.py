"""Whole configs[2] frame at 256 spp with forced sample-chunk counts (0 = the library's choice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from spath_amd import capi, scene, view
ctx = capi.Context(0)
t, m = scene.closed_room(10000)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
w, h, spp = 1920, 1080, 256
d_t, d_m, d_r = d(t), d(m), d(view.Camera(w, h).get_viewport())
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), 10000, 0)
out = torch.zeros(w * h, 4, dtype=torch.uint8, device="cuda")
for ch in (0, 128, 64, 32, 16, 8, 4, 1, 0):
    ctx.render_device(d_r.data_ptr(), w * h, spp, out.data_ptr(), flags=capi.flag_chunks(ch)); torch.cuda.synchronize()
    st = ctx.stats()
    print(f"chunks {ch:3d}: {st['kernel_ms']:8.1f} ms  {w*h*spp*5/st['kernel_ms']/1e3:7.1f} Mray/s  launches {st['n_launches']}  image sum {int(out.sum())}", flush=True)
