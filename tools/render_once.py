"""One path-traced frame with a given kernel variant (for rocprofv3 --pmc passes and stats builds).
python tools/render_once.py <variant name|id> [w h spp tris [lib.so]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi
if len(sys.argv) > 6: capi.LIB_PATH = os.path.abspath(sys.argv[6])
from spath_amd import scene, view
var = sys.argv[1]
w, h, spp, nt = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 16, 10000)))
flags = capi.kernel_variants()[var] if not var.isdigit() else int(var)
ctx = capi.Context(0)
t, m = scene.closed_room(nt)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
d_t, d_m, d_r = d(t), d(m), d(view.Camera(w, h).get_viewport())
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), nt, 0)
out = torch.zeros(w * h, 4, dtype=torch.uint8, device="cuda")
for rep in range(int(os.environ.get("REPS", "1"))):
    ctx.render_device(d_r.data_ptr(), w * h, spp, out.data_ptr(), flags=flags)
    torch.cuda.synchronize()
    st = ctx.stats()
    print(f"{var} {w}x{h}x{spp} {nt} tris: kernel {st['kernel_ms']:.2f} ms, {w*h*spp*5/st['kernel_ms']/1e3:.1f} Mray/s, {st['scans_executed']*nt/st['kernel_ms']/1e9:.3f} T tests/s, scans {st['scans_executed']}, image sum {int(out.sum())}", flush=True)
