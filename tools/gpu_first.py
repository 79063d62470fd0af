"""First contact with the GPU: parity spot checks + a timing + the VALU micro-benchmark."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
from oracle import oracle as O
from spath_amd import capi, scene, view

ctx = capi.Context(0)
print(ctx.description, capi.kernel_variants())

def check(name, tris, mats, w, h, spp, seed=1):
  for var in (1, 2):
    rays = view.Camera(w, h).get_viewport()
    ctx.set_scene(tris, mats)
    flat = ctx.render(rays, w, h, 1, mode=capi.MODE_FLAT, flags=var)
    oflat = O.render_flat(rays, w, h, tris, mats)
    img, acc = ctx.render(rays, w, h, spp, seed=seed, want_accum=True, flags=var)
    st = ctx.stats()
    t0 = time.time(); oimg, oacc, scans = O.render_counter(rays, tris, mats, spp, seed); t1 = time.time()
    nb = int((img != oimg).any(axis=1).sum()); na = int((acc != oacc).any(axis=1).sum())
    print(f"{name} v{var}: flat_equal={np.array_equal(flat, oflat)} pt_rgba_mismatch_px={nb} accum_mismatch_px={na} "
          f"scans gpu={st['scans_executed']} cpu={scans} kernel_ms={st['kernel_ms']:.2f} cpu_s={t1-t0:.2f}", flush=True)

t, m = scene.default_scene()
check("default 320x240x4", t, m, 320, 240, 4)
t, m = scene.closed_room(1000)
check("closed1k 96x64x2", t, m, 96, 64, 2)
t, m = scene.open_clutter(300)
check("open300 128x96x3", t, m, 128, 96, 3, seed=0xDEADBEEF12345)

t, m = scene.closed_room(10000)
ctx.set_scene(t, m)
for (w, h, spp, var) in ((960, 540, 4, 1), (960, 540, 4, 2), (1920, 1080, 4, 1), (1920, 1080, 4, 2), (1920, 1080, 8, 2)):
    rays = view.Camera(w, h).get_viewport()
    ctx.render(rays, w, h, spp, flags=var)
    st = ctx.stats()
    mray = w * h * spp * 5 / (st['kernel_ms'] * 1e-3) / 1e6
    print(f"closed10k variant={var} {w}x{h}x{spp}: kernel {st['kernel_ms']:.1f} ms scans={st['scans_executed']} nominal={w*h*spp*5} -> {mray:.1f} Mray/s, "
          f"{st['scans_executed']*10000/(st['kernel_ms']*1e-3)/1e12:.3f} T tests/s", flush=True)
ctx.close()
vb = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "rcp_check")
if os.path.exists(vb):
    print(subprocess.run([vb], capture_output=True, text=True).stdout)
