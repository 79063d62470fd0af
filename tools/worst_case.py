"""The large-triangle scene of bench.py's worst_case_untimed (closed_room(N, clutter_scale=10)) by kernel variant: python tools/worst_case.py [spp]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene, view
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w, h, nt = 1920, 1080, 10000
ctx = capi.Context(0)
t, m = scene.closed_room(nt, clutter_scale=10.0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
d_t, d_m, d_r = d(t), d(m), d(view.Camera(w, h).get_viewport())
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), nt, 0)
names = {v: k for k, v in capi.kernel_variants().items()}
for var in (16, 15, 12, 2):
    out = torch.zeros(w * h, 4, dtype=torch.uint8, device="cuda")
    for rep in range(2):
        ctx.render_device(d_r.data_ptr(), w * h, spp, out.data_ptr(), seed=1, flags=var); torch.cuda.synchronize()
    st = ctx.stats()
    print(f"{names[var]:11s} {w}x{h}x{spp} clutter x10: {st['kernel_ms']:.1f} ms, {w*h*spp*5/st['kernel_ms']/1e3:.1f} Mray/s, {st['scans_executed']*nt/st['kernel_ms']/1e9:.3f} T tests/s, "
          f"rgba sha256 {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
