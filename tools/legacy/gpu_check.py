"""Quick GPU spot check: parity of every kernel variant vs the oracle + timings on the closed-room scene."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from spath_amd import capi, scene, view

ctx = capi.Context(0)
V = capi.kernel_variants()
print(ctx.description, V, flush=True)
variants = [v for n, v in V.items() if v > 0]
if len(sys.argv) > 1:
    variants = [int(x) for x in sys.argv[1].split(",")]

def check(name, tris, mats, w, h, spp, seed=1):
    rays = view.Camera(w, h).get_viewport()
    ctx.set_scene(tris, mats)
    oflat = O.render_flat(rays, w, h, tris, mats)
    oimg, oacc, scans = O.render_counter(rays, tris, mats, spp, seed)
    for var in variants:
        flat = ctx.render(rays, w, h, 1, mode=capi.MODE_FLAT, flags=var)
        img, acc = ctx.render(rays, w, h, spp, seed=seed, want_accum=True, flags=var)
        st = ctx.stats()
        nb = int((img != oimg).any(axis=1).sum()); na = int((acc != oacc).any(axis=1).sum())
        nf = int((flat != oflat).any(axis=1).sum())
        ok = (nb == 0 and na == 0 and nf == 0 and st['scans_executed'] == scans)
        print(f"{'OK  ' if ok else 'FAIL'} {name} v{var}: flat_mismatch={nf} rgba_mismatch={nb} accum_mismatch={na} "
              f"scans gpu={st['scans_executed']} cpu={scans} kernel_ms={st['kernel_ms']:.2f}", flush=True)

t, m = scene.default_scene()
check("default 320x240x4", t, m, 320, 240, 4)
check("default 67x41x3", t, m, 67, 41, 3, seed=99)
t, m = scene.closed_room(1000)
check("closed1k 96x64x2", t, m, 96, 64, 2)
t, m = scene.open_clutter(300)
check("open300 128x96x3", t, m, 128, 96, 3, seed=0xDEADBEEF12345)
t, m = scene.closed_room(3000, clutter_scale=8.0)   # large triangles: many slab survivors, queue overflow path
check("closed3k-big 64x48x2", t, m, 64, 48, 2)

t, m = scene.closed_room(10000)
ctx.set_scene(t, m)
for (w, h, spp) in ((1920, 1080, 4),):
    rays = view.Camera(w, h).get_viewport()
    for var in variants:
        ctx.render(rays, w, h, spp, flags=var)
        st = ctx.stats()
        mray = w * h * spp * 5 / (st['kernel_ms'] * 1e-3) / 1e6
        print(f"closed10k variant={var} {w}x{h}x{spp}: kernel {st['kernel_ms']:.1f} ms scans={st['scans_executed']} -> {mray:.1f} Mray/s, "
              f"{st['scans_executed']*10000/(st['kernel_ms']*1e-3)/1e12:.3f} T tests/s", flush=True)
ctx.close()
