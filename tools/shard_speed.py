"""Speed of one rank's shard of a 1080p frame for 1/2/4/8-way row-tile splits (SPP env, default 8 spp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene, view
from spath_amd.dist import RowTilePlan, ShardedRenderer
ctx = capi.Context(0)
t, m = scene.closed_room(10000)
d_t, d_m = torch.from_numpy(t).cuda(), torch.from_numpy(m).cuda()
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), 10000, 0)
rays = view.Camera(1920, 1080).get_viewport()
spp = int(os.environ.get("SPP", "8"))
for world in (8, 4, 2, 1):
    plan = RowTilePlan(1920, 1080, world, 8)
    sh = ShardedRenderer(ctx, plan, 0, rays, torch.device("cuda"))
    for var in (0, 6):
        for rep in range(2):
            sh.render(spp, flags=var); torch.cuda.synchronize(); st = ctx.stats()
        print(f"world {world}: {spp} spp, shard {sh.n} px, variant {var}->{st['kernel_variant']}: {st['kernel_ms']:.1f} ms, {st['scans_executed']*1e4/st['kernel_ms']/1e9:.3f} T tests/s", flush=True)
