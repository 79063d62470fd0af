"""render_flat (one scan per pixel) timing by kernel variant: python tools/flat_time.py [w h tris]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from spath_amd import capi, scene, view
w, h, nt = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1920, 1080, 10000)))
names = {v: k for k, v in capi.kernel_variants().items()}
ctx = capi.Context(0)
t, m = scene.closed_room(nt)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
d_t, d_m, d_r = d(t), d(m), d(view.Camera(w, h).get_viewport())
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), nt, 0)
out = torch.zeros(w * h, 4, dtype=torch.uint8, device="cuda")
ref = None
for var in (0, 16, 14, 11, 10, 9, 3, 2):
    best = 1e9
    for rep in range(3):
        ctx.render_device(d_r.data_ptr(), w * h, 1, out.data_ptr(), mode=capi.MODE_FLAT, flags=var); torch.cuda.synchronize()
        st = ctx.stats(); best = min(best, st["kernel_ms"])
    img = out.cpu().numpy()
    if ref is None: ref = img.copy()
    print(f"flat {w}x{h} {nt} tris  {names[var]:12s} -> {names[st['kernel_variant']]:12s} {best:8.2f} ms  {w*h*nt/best/1e9:.3f} T tests/s  image {'same' if np.array_equal(img, ref) else 'DIFFERS'}", flush=True)
