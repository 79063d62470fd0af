"""A/B of alternative builds (build/abl_*.so): worst-case scene (clutter x10) and the 100k-triangle 4K slice: python tools/ab_big.py"""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = """
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from spath_amd import capi
capi.LIB_PATH = sys.argv[1]
from spath_amd import scene, view
ctx = capi.Context(0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for tag, (t, m), w, h, spp in (("clutter x10, 10k, 1080p x 4 spp", scene.closed_room(10000, clutter_scale=10.0), 1920, 1080, 4),
                               ("100k tris, 4K rows 0-269 x 8 spp", scene.closed_room(100000), 3840, 270, 8),
                               ("1M tris, 4K rows 0-63 x 4 spp", scene.closed_room(1000000), 3840, 64, 4)):
    nt = t.shape[0]
    rays = view.Camera(3840 if w == 3840 else w, 2160 if w == 3840 else h).get_viewport().reshape(-1, 6)[: w * h]
    d_t, d_m, d_r = d(t), d(m), d(rays)
    ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), nt, 0)
    out = torch.zeros(w * h, 4, dtype=torch.uint8, device='cuda')
    best = 1e30
    for rep in range(2):
        ctx.render_device(d_r.data_ptr(), w * h, spp, out.data_ptr(), seed=1, flags=0); torch.cuda.synchronize()
        st = ctx.stats(); best = min(best, st['kernel_ms'])
    print(f"{tag}: {best:.1f} ms, {st['scans_executed']*nt/best/1e9:.3f} T tests/s, image sum {int(out.sum())}")
""" % root
for lib in sorted(glob.glob(os.path.join(root, "build", "abl_*.so"))):
    p = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True)
    for l in p.stdout.strip().splitlines(): print(f"{os.path.basename(lib):18s} {l}", flush=True)
    if p.returncode: print("FAILED", p.stderr[-400:], flush=True)
