"""configs[2] at 256 spp with the number of sample chunks forced (SPHIP_FLAG_CHUNKS): kernel time by chunk count, same image.
usage: python tools/chunk_ab.py [chunk counts...]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene, view

W, H, SPP, NT = 1920, 1080, 256, 10000
counts = [int(x) for x in sys.argv[1:]] or [0, 8, 16, 32, 64, 128]
ctx = capi.Context(0)
t, m = scene.closed_room(NT)
rays = view.Camera(W, H).get_viewport()
dev = torch.device("cuda:0")
d_t, d_m, d_r = torch.from_numpy(t).to(dev), torch.from_numpy(m).to(dev), torch.from_numpy(rays).to(dev)
out = torch.zeros(W * H, 4, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), NT, st)
for c in counts:
    for rep in range(2):
        ctx.render_device(d_r.data_ptr(), W * H, SPP, out.data_ptr(), seed=1, mode=capi.MODE_PT, flags=c << 16, stream=st)
        torch.cuda.synchronize()
    s = ctx.stats()
    print(f"chunks {c or 'auto':>4}: kernel {s['kernel_ms']:.1f} ms, {W * H * SPP * 5 / s['kernel_ms'] / 1e3:.1f} Mray/s, sha {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
