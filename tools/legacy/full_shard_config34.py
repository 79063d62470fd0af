"""configs[3] at its REAL size: rank 0's whole shard of the 8-GPU row-tile plan (3840x2160, 100k triangles, 1024 spp: 5.3e14
ray-triangle pairs) through the exact-only scan and the default two-stage scan, piece by piece (groups of 4 row tiles, so that the
run reports progress): RGBA8 and float accumulators must be identical.  Then the configs[4] slice (1M triangles, 64 rows x 64 spp).
python tools/full_shard_config34.py [spp3 [spp4 [3|4|34]]]      ONLY_TWO_STAGE=1: skip the exact-only scan (18 minutes at configs[3]) and
print the two-stage SHA-256 to compare with the exact-only one of an earlier run (same seeds, same pieces: same bytes)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from spath_amd import capi, scene, view
from spath_amd.dist import RowTilePlan

spp3 = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp4 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = capi.Context(0)
W, H = 3840, 2160
rays = view.Camera(W, H).get_viewport().reshape(-1, 6)
dev = torch.device("cuda")
ok_all = True


def run(tag, ntri, spp, tiles_per_piece, n_tiles_limit=None):
    global ok_all
    t, m = scene.closed_room(ntri)
    d_t, d_m = torch.from_numpy(t).to(dev), torch.from_numpy(m).to(dev)
    ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), ntri, 0)
    plan = RowTilePlan(W, H, 8, 8)
    tiles = plan.tiles_of(0)[: n_tiles_limit]
    h_all = {"two-stage": hashlib.sha256(), "exact": hashlib.sha256()}
    t_ms = {"two-stage": 0.0, "exact": 0.0}
    for j0 in range(0, len(tiles), tiles_per_piece):
        part = tiles[j0:j0 + tiles_per_piece]
        ids = np.concatenate([np.arange(tl * plan.tile_px, min((tl + 1) * plan.tile_px, plan.npix), dtype=np.int64) for tl in part])
        n = ids.size
        d_r = torch.from_numpy(np.ascontiguousarray(rays[ids])).to(dev)
        shard = (part[0] * plan.tile_px, plan.tile_px, 8 * plan.tile_px)
        res = {}
        for name, fl in ((("two-stage", 0),) if os.environ.get("ONLY_TWO_STAGE") else (("two-stage", 0), ("exact", 2))):
            img = torch.zeros(n, 4, dtype=torch.uint8, device=dev); acc = torch.zeros(n, 3, dtype=torch.float32, device=dev)
            ctx.render_device(d_r.data_ptr(), n, spp, img.data_ptr(), seed=1, flags=fl, shard=shard, image_width=W, d_out_accum=acc.data_ptr())
            torch.cuda.synchronize(); st = ctx.stats()
            res[name] = (img.cpu().numpy(), acc.cpu().numpy(), st["scans_executed"])
            h_all[name].update(res[name][0].tobytes()); h_all[name].update(res[name][1].tobytes())
            t_ms[name] += st["kernel_ms"]
        if "exact" not in res: res["exact"] = res["two-stage"]          # ONLY_TWO_STAGE=1: compare the final SHA-256 with a recorded exact-only run
        same = np.array_equal(res["two-stage"][0], res["exact"][0]) and np.array_equal(res["two-stage"][1], res["exact"][1]) and res["two-stage"][2] == res["exact"][2]
        ok_all &= same
        print(f"{tag}: tiles {part[0]}..{part[-1]} ({n} px x {spp} spp): {'identical' if same else 'DIFFERENT'}; two-stage {t_ms['two-stage']/1e3:.1f} s, exact {t_ms['exact']/1e3:.1f} s so far", flush=True)
    print(f"{tag}: sha256 two-stage {h_all['two-stage'].hexdigest()[:16]} exact {h_all['exact'].hexdigest()[:16]}; kernel time two-stage {t_ms['two-stage']/1e3:.1f} s, exact-only {t_ms['exact']/1e3:.1f} s", flush=True)


which = sys.argv[3] if len(sys.argv) > 3 else "34"
if "3" in which: run("configs[3] rank-0 shard (100k tris, 4K, %d spp)" % spp3, 100000, spp3, 4)
if "4" in which: run("configs[4] slice (1M tris, 4K, 64 rows x %d spp)" % spp4, 1000000, spp4, 2, n_tiles_limit=8)
print("ALL IDENTICAL" if ok_all else "MISMATCH", flush=True)
sys.exit(0 if ok_all else 1)
