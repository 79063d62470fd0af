"""Declared slices of BASELINE.json configs[3] and configs[4] on ONE MI355X.

configs[3]: 100k triangles, 3840x2160, 1024 spp, pixel-row tiles over 8 GPUs -> this runs rank 0's shard of the
            8-way interleaved 8-row-tile plan at FULL spp (what each of the 8 GPUs does) and projects the 8-GPU rate.
configs[4]: 1M triangles, 3840x2160, 4096 spp -> 64 interleaved rows x 64 spp, rate extrapolated.
"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from spath_amd import capi, scene, view
from spath_amd.dist import RowTilePlan, ShardedRenderer

ctx = capi.Context(0)
dev = torch.device("cuda")
W, H = 3840, 2160
rays = view.Camera(W, H).get_viewport()
out = {}

def run(tag, ntri, plan, rank, spp, full_spp):
    t, m = scene.closed_room(ntri)
    d_t, d_m = torch.from_numpy(t).to(dev), torch.from_numpy(m).to(dev)
    ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), ntri, 0)
    sh = ShardedRenderer(ctx, plan, rank, rays, dev)
    t0 = time.time()
    sh.render(spp); torch.cuda.synchronize(); st = ctx.stats()
    scans = st["scans_executed"]; ms = st["kernel_ms"]
    nominal_slice = sh.n * spp * 5
    rate = nominal_slice / (ms * 1e-3) / 1e6
    out[tag] = {"n_tris": ntri, "slice_pixels": sh.n, "slice_spp": spp, "kernel_ms": ms, "scans_executed": scans, "nominal_scans_slice": nominal_slice,
                "Mray_per_s_this_gpu": rate, "T_tests_per_s": scans * ntri / (ms * 1e-3) / 1e12,
                "effective_GBps": scans * ntri * 48 / (ms * 1e-3) / 1e9, "kernel": ctx._L.sphip_kernel_name(st["kernel_variant"]).decode(),
                "full_config_seconds_on_this_rate": W * H * full_spp * 5 / (rate * 1e6)}
    print(tag, json.dumps(out[tag]), flush=True)

run("configs[3] rank-0 shard of 8 (100k tris, 4K, 1024 spp)", 100000, RowTilePlan(W, H, 8, 8), 0, 1024, 1024)
run("configs[4] slice (1M tris, 4K, 64 rows x 64 spp of 4096)", 1000000, RowTilePlan(W, H, 34, 8), 0, 64, 4096)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "config45.json"), "w"), indent=1)
