"""Rate of the whole-frame path-traced launch vs samples per pixel and iterations per lane (sample chunks forced)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from spath_amd import capi, scene, view
from spath_amd.dist import RowTilePlan, ShardedRenderer
ctx = capi.Context(0)
t, m = scene.closed_room(10000)
d_t, d_m = torch.from_numpy(t).cuda(), torch.from_numpy(m).cuda()
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), 10000, 0)
rays = view.Camera(1920, 1080).get_viewport()
sh = ShardedRenderer(ctx, RowTilePlan(1920, 1080, 1, 8), 0, rays, torch.device("cuda"))
for spp in (8, 16, 32, 64):
    n_iter = spp // 2
    for ipl in (1, 2, 4, 8):
        ch = n_iter // ipl
        if ch < 1 or ch > 255:
            continue
        best = 1e9
        for rep in range(2):
            sh.render(spp, flags=capi.flag_chunks(ch)); torch.cuda.synchronize(); st = ctx.stats(); best = min(best, st["kernel_ms"])
        print(f"{spp:3d} spp, chunks {ch:3d} ({ipl} iterations per lane): {best:8.1f} ms  {st['scans_executed']*1e4/best/1e9:.3f} T tests/s", flush=True)
