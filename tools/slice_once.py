"""One declared slice of BASELINE.json configs[3] or configs[4] on one MI355X, for rocprofv3 passes (kernel trace, PMC):
  python3 tools/slice_once.py config3 [spp]     100k triangles, 3840x2160: rank 0's shard of the 8-GPU row-tile plan (full config: 1024 spp)
  python3 tools/slice_once.py config4 [spp]     1M triangles, 3840x2160: 64 interleaved rows (rank 0 of 34) (full config: 4096 spp)
Prints one JSON line (scans, kernel ms, tests/s, library source hash) that tools/slice_profiles.py pairs with the counters."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene, view
from spath_amd.dist import RowTilePlan, ShardedRenderer

which = sys.argv[1]
W, H = 3840, 2160
ntri, plan, full_spp = (100000, RowTilePlan(W, H, 8, 8), 1024) if which == "config3" else (1000000, RowTilePlan(W, H, 34, 8), 4096)
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = capi.Context(0)
dev = torch.device("cuda")
rays = view.Camera(W, H).get_viewport()
t, m = scene.closed_room(ntri)
d_t, d_m = torch.from_numpy(t).to(dev), torch.from_numpy(m).to(dev)
ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), ntri, 0)
sh = ShardedRenderer(ctx, plan, 0, rays, dev)
sh.render(1); torch.cuda.synchronize()                     # builds the record stream (the profiled launch below is the render alone)
sh.render(spp); torch.cuda.synchronize()
st = ctx.stats()
kname = capi.load().sphip_kernel_name(st["kernel_variant"]).decode()
print(json.dumps({"slice": which, "n_tris": ntri, "width": W, "height": H, "slice_pixels": sh.n, "spp": spp, "full_config_spp": full_spp, "kernel": kname,
                  "kernel_ms": st["kernel_ms"], "scans_executed": st["scans_executed"], "tests": st["scans_executed"] * ntri,
                  "T_tests_per_s": st["scans_executed"] * ntri / (st["kernel_ms"] * 1e-3) / 1e12,
                  "Mray_per_s_this_gpu": sh.n * spp * 5 / (st["kernel_ms"] * 1e-3) / 1e6,
                  "library_source_hash": capi.build_source_hash(),
                  "key": f"{ntri}tris_{W}x{H}_slice{sh.n}px_x{spp}spp_{kname}"}), flush=True)
