"""SPHIP_FLAG_PRIMARY_REUSE on the configs[2] frame: time and SHA-256 with and without the flag.
python tools/primary_reuse_time.py [spp [tris w h]]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene, view
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nt, w, h = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (10000, 1920, 1080)
ctx = capi.Context(0)
t, m = scene.closed_room(nt)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
d_t, d_m, d_r = d(t), d(m), d(view.Camera(w, h).get_viewport())
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), nt, 0)
for name, flags in (("default", 0), ("primary reuse", capi.FLAG_PRIMARY_REUSE), ("default", 0), ("primary reuse", capi.FLAG_PRIMARY_REUSE)):
    out = torch.zeros(w * h, 4, dtype=torch.uint8, device="cuda"); acc = torch.zeros(w * h, 3, dtype=torch.float32, device="cuda")
    ctx.render_device(d_r.data_ptr(), w * h, spp, out.data_ptr(), seed=1, flags=flags, d_out_accum=acc.data_ptr())
    torch.cuda.synchronize()
    st = ctx.stats()
    print(f"{name:14s} {w}x{h}x{spp} {nt} tris: {st['kernel_ms']:.1f} ms, {w*h*spp*5/st['kernel_ms']/1e3:.1f} nominal Mray/s, scans executed {st['scans_executed']} "
          f"({st['scans_executed']*nt/st['kernel_ms']/1e9:.3f} T tests/s executed), launches {st['n_launches']}, kernel variant {st['kernel_variant']}, "
          f"rgba sha256 {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]} accum sha256 {hashlib.sha256(acc.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
