"""Sample-chunk sweep: kernel time of the path-traced launch for forced chunk counts, whole 1080p frame and 1/8 shard."""
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
spp = int(os.environ.get("SPP", "32"))
for world in (1, 2, 8):
    sh = ShardedRenderer(ctx, RowTilePlan(1920, 1080, world, 8), 0, rays, torch.device("cuda"))
    for ch in (1, 0, 2, 4, 8, 16):
        best = 1e9
        for rep in range(2):
            sh.render(spp, flags=capi.flag_chunks(ch)); torch.cuda.synchronize(); st = ctx.stats(); best = min(best, st["kernel_ms"])
        print(f"1/{world} frame, {spp} spp, chunks {ch:2d}: {best:8.1f} ms  {st['scans_executed']*1e4/best/1e9:.3f} T tests/s  launches {st['n_launches']}", flush=True)
