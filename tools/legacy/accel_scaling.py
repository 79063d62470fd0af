"""SURVEY 8(f4) opt-in acceleration structure: nominal Mray/s and agreement with the brute-force scan by scene size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spath_amd import capi, scene, view
ctx = capi.Context(0)
for ntri, (w, h, spp) in ((10000, (1920, 1080, 8)), (100000, (1920, 1080, 2)), (1000000, (960, 540, 1))):
    t, m = scene.closed_room(ntri)
    ctx.set_scene(t, m)
    rays = view.Camera(w, h).get_viewport()
    t0 = time.time(); ctx.render(rays, 8, 1, 1, flags=capi.FLAG_ACCEL) if False else None
    res = {}
    for name, fl in (("brute force (rpl_filter2s)", 0), ("accel (lbvh)", capi.FLAG_ACCEL)):
        t0 = time.time()
        img, acc = ctx.render(rays, w, h, spp, flags=fl, want_accum=True); st = ctx.stats()
        wall = time.time() - t0
        res[name] = (img, acc)
        print(f"{ntri:8d} tris {w}x{h}x{spp} {name:28s}: kernel {st['kernel_ms']:9.1f} ms (call {wall:.2f} s incl. build/transfers), "
              f"{w*h*spp*5/st['kernel_ms']/1e3:9.1f} nominal Mray/s", flush=True)
    a, b = res["brute force (rpl_filter2s)"], res["accel (lbvh)"]
    print(f"         pixels whose accumulator differs: {(a[1] != b[1]).any(axis=1).sum()} of {w*h}; RGBA8 differs: {(a[0] != b[0]).any(axis=1).sum()}", flush=True)
