"""A/B of alternative builds (build/abl_*.so) on the flat pass (one scan per pixel, all rays from the camera): python tools/flat_ab.py [w h tris]"""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:4] if len(sys.argv) > 3 else ["3840", "2160", "10000"]
code = """
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from spath_amd import capi
capi.LIB_PATH = sys.argv[4]
from spath_amd import scene, view
w, h, nt = (int(x) for x in sys.argv[1:4])
ctx = capi.Context(0)
t, m = scene.closed_room(nt)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
d_t, d_m, d_r = d(t), d(m), d(view.Camera(w, h).get_viewport())
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), nt, 0)
out = torch.zeros(w * h, 4, dtype=torch.uint8, device='cuda')
best = 1e9
for rep in range(4):
    ctx.render_device(d_r.data_ptr(), w * h, 1, out.data_ptr(), mode=capi.MODE_FLAT, flags=14); torch.cuda.synchronize()
    best = min(best, ctx.stats()['kernel_ms'])
print(f'flat {w}x{h} {nt} tris rpl_cylw4: {best:.2f} ms, {w*h*nt/best/1e9:.3f} T tests/s, image sum {int(out.sum())}')
""" % root
for lib in sorted(glob.glob(os.path.join(root, "build", "abl_*.so"))):
    p = subprocess.run([sys.executable, "-c", code] + args + [lib], capture_output=True, text=True)
    print(f"{os.path.basename(lib):18s} {p.stdout.strip().splitlines()[-1] if p.stdout.strip() else 'FAILED ' + p.stderr[-300:]}", flush=True)
