"""Configs 4 and 5 shapes (100k and 1M triangles) at reduced image size: throughput + exact-vs-filter agreement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spath_amd import capi, scene, view
ctx = capi.Context(0)
for ntri, (w, h, spp) in ((100000, (960, 540, 1)), (1000000, (480, 270, 1))):
    t, m = scene.closed_room(ntri)
    ctx.set_scene(t, m)
    rays = view.Camera(w, h).get_viewport()
    out = {}
    for var in (2, 3):
        img, acc = ctx.render(rays, w, h, spp, flags=var, want_accum=True); st = ctx.stats()
        out[var] = (img, acc)
        print(f"{ntri} tris {w}x{h}x{spp} variant {var}: {st['kernel_ms']:.1f} ms, scans {st['scans_executed']} (nominal {w*h*spp*5}), "
              f"{st['scans_executed']*ntri/st['kernel_ms']/1e9:.3f} T tests/s, {w*h*spp*5/st['kernel_ms']/1e3:.2f} Mray/s", flush=True)
    print("   exact == filter:", np.array_equal(out[2][0], out[3][0]) and np.array_equal(out[2][1], out[3][1]), flush=True)
