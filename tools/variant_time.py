"""Time the path-trace kernel of each given libspath_hip build on the whole 1080p frame (8 spp) and on an 8-way shard."""
import os, sys, glob, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    import numpy as np, torch
    from spath_amd import capi
    capi.LIB_PATH = sys.argv[2]
    from spath_amd import scene, view
    from spath_amd.dist import RowTilePlan, ShardedRenderer
    ctx = capi.Context(0)
    t, m = scene.closed_room(10000)
    d_t, d_m = torch.from_numpy(t).cuda(), torch.from_numpy(m).cuda()
    ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), 10000, 0)
    rays = view.Camera(1920, 1080).get_viewport()
    out = []
    for world in (1, 8):
        sh = ShardedRenderer(ctx, RowTilePlan(1920, 1080, world, 8), 0, rays, torch.device("cuda"))
        best = 1e9
        for rep in range(3):
            sh.render(8, flags=int(os.environ.get("VAR", "0"))); torch.cuda.synchronize(); st = ctx.stats(); best = min(best, st["kernel_ms"])
        out.append(f"1/{world} frame: {best:7.1f} ms {st['scans_executed']*1e4/best/1e9:.3f} T tests/s")
    print(f"{os.path.basename(sys.argv[2]):28s}", " | ".join(out), flush=True)
else:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for lib in sorted(glob.glob(os.path.join(root, "build", "abl_*.so"))):
        subprocess.run([sys.executable, __file__, "--one", lib])
