"""Where the 512-thread shape (octet bits) overtakes the 256-thread one: closed rooms of growing size, both shapes forced in turn
(SPATH_HIP_CYLM_SHAPE is read when the stream is built: set before every set_scene).  1920 x 270 x 8 spp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene, view

W, H, SPP = 1920, 270, 8
ctx = capi.Context(0)
rays = view.Camera(W, 1080).get_viewport()[: W * H]
dev = torch.device("cuda:0")
d_r = torch.from_numpy(np.ascontiguousarray(rays)).to(dev)
out = torch.zeros(W * H, 4, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for n in [int(x) for x in sys.argv[1:]] or [8192, 12288, 16384, 24576, 32768, 49152, 65536]:
    t, m = scene.closed_room(n)
    d_t, d_m = torch.from_numpy(t).to(dev), torch.from_numpy(m).to(dev)
    res = {}
    for shape in ("256", "512"):
        os.environ["SPATH_HIP_CYLM_SHAPE"] = shape
        ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), n, st)
        for rep in range(2):
            ctx.render_device(d_r.data_ptr(), W * H, SPP, out.data_ptr(), seed=1, mode=capi.MODE_PT, stream=st)
            torch.cuda.synchronize()
        s = ctx.stats()
        res[shape] = (s["kernel_ms"], int(out.to(torch.int64).sum()))
    assert res["256"][1] == res["512"][1]
    print(f"{n:6d} triangles: 256-thread shape {res['256'][0]:8.2f} ms, 512-thread shape {res['512'][0]:8.2f} ms -> {'512' if res['512'][0] < res['256'][0] else '256'} ({res['256'][0] / res['512'][0]:.3f})", flush=True)
