"""Where a wave spends its life inside the default scan (diagnostic build -DSP_PHASE_TIMERS: s_memtime stamps at the phase
boundaries, summed per wave -> stderr of sphip_get_stats): python tools/phase_timers.py [lib.so ...]   (default build/phase_timers*.so)
The stamps cost ~10 %: read the split, not the total."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "build", "phase_timers*.so")))
code = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from spath_amd import capi
capi.LIB_PATH = sys.argv[1]
from spath_amd import scene, view
ctx = capi.Context(0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
r4 = view.Camera(3840, 2160).get_viewport().reshape(-1, 6)
for tag, (t, m), rays, w, h, spp in (("configs[2] 10k 1080p x 8 spp", scene.closed_room(10000), view.Camera(1920, 1080).get_viewport(), 1920, 1080, 8),
                                     ("clutter x10 10k 1080p x 2 spp", scene.closed_room(10000, clutter_scale=10.0), view.Camera(1920, 1080).get_viewport(), 1920, 1080, 2),
                                     ("100k tris, 4K rows 0-269 x 4 spp", scene.closed_room(100000), r4[:3840 * 270], 3840, 270, 4)):
    nt = t.shape[0]
    d_t, d_m, d_r = d(t), d(m), d(rays)
    ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), nt, 0)
    out = torch.zeros(w * h, 4, dtype=torch.uint8, device='cuda')
    for rep in range(2):
        ctx.render_device(d_r.data_ptr(), w * h, spp, out.data_ptr(), seed=1, flags=0); torch.cuda.synchronize()
    sys.stderr.write(tag + ": "); sys.stderr.flush()
    st = ctx.stats()
    print(f"{tag}: {st['kernel_ms']:.1f} ms, {st['scans_executed']*nt/st['kernel_ms']/1e9:.3f} T tests/s", flush=True)
""" % root
for lib in libs:
    p = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True)
    print(f"== {os.path.basename(lib)}"); print(p.stdout.strip()); print(p.stderr.strip()[-3000:], flush=True)
