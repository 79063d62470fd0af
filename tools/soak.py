"""Randomized soak of the two-stage scans against the exact-only scan (bit for bit): python tools/soak.py [seconds] [seed]
Random scenes (closed room, open clutter, raw triangle soups over several decades of size with slivers, duplicated and degenerate
triangles; triangle counts around the 64 / 256 / 384 / 512-triangle boundaries), random image sizes, spp, seeds, shards and chunk
counts; every third scene forces the other workgroup shape of the default scan (SPATH_HIP_CYLM_SHAPE).
Checks path-traced accumulators + RGBA8 + scan counts, the flat pass, and the scan alone with random idx_source."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene, view

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
ctx = capi.Context(0)
TWO = [v for v in capi.available_variants() if v >= 3 and v != 8]        # shipped build: 15, 16; -DSP_ALL_VARIANTS: every generation
names = {v: k for k, v in capi.kernel_variants().items()}


def soup(n):
    scale = 10.0 ** rng.uniform(-2.5, 0.5, (n, 1))
    ctr = rng.uniform(-2, 2, (n, 3)) * [1, 0.6, 1]
    t = np.zeros((n, 12), dtype=np.float32)
    for k in range(3):
        t[:, 3 * k:3 * k + 3] = ctr + rng.normal(size=(n, 3)) * scale
    sl = rng.random(n) < 0.1                                  # slivers: third vertex almost on the first edge
    t[sl, 6:9] = t[sl, 0:3] + (t[sl, 3:6] - t[sl, 0:3]) * rng.uniform(0, 1, (int(sl.sum()), 1)) + rng.normal(size=(int(sl.sum()), 3)) * 1e-5
    dg = rng.random(n) < 0.02                                 # degenerate: two equal vertices
    t[dg, 3:6] = t[dg, 0:3]
    if n > 4:
        du = rng.integers(0, n, max(1, n // 50)); t[du] = t[rng.integers(0, n, du.size)]       # duplicates: ties -> lowest index
    t = scene.flat_normals(t)
    t[:, 9:12] = np.nan_to_num(t[:, 9:12])
    m = np.zeros((n, 6), dtype=np.float32); m[:, :3] = rng.uniform(0.1, 1.0, (n, 3)); m[rng.random(n) < 0.05, 3:] = 1.0
    return t, m


t_end, it, fails = time.time() + budget, 0, 0
while time.time() < t_end:
    it += 1
    n = int(rng.choice([rng.integers(1, 70), rng.integers(250, 262), rng.integers(378, 392), rng.integers(506, 520), rng.integers(760, 776), rng.integers(1020, 1030),
                        rng.integers(64, 4000), rng.integers(4000, 30000), rng.integers(30000, 60000)]))
    if it % 3 == 0: os.environ["SPATH_HIP_CYLM_SHAPE"] = str(rng.choice([256, 512]))
    else: os.environ.pop("SPATH_HIP_CYLM_SHAPE", None)
    kind = rng.integers(0, 3)
    if kind == 0 and n >= 14: t, m = scene.closed_room(n, seed=int(rng.integers(1, 1 << 30)), clutter_scale=float(rng.choice([1.0, 1.0, 3.0, 10.0])))
    elif kind == 1 and n >= 7: t, m = scene.open_clutter(n, seed=int(rng.integers(1, 1 << 30)))
    else: t, m = soup(n)
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    spp, seed = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 17])), int(rng.integers(0, 1 << 62))
    cam = view.Camera(w, h)
    cam.set_delta_mov(tuple(rng.uniform(-0.5, 0.5, 3))); cam.set_delta_rot(tuple(rng.uniform(-0.4, 0.4, 3)))
    rays = cam.get_viewport()
    ctx.set_scene(t, m)
    chunks = capi.flag_chunks(int(rng.choice([0, 0, 1, 2, 7])))
    want = ctx.render(rays, w, h, spp, seed=seed, flags=2, want_accum=True); ws = ctx.stats()["scans_executed"]
    wflat = ctx.render(rays, w, h, 1, mode=capi.MODE_FLAT, flags=2)
    nr = 20000
    hr = np.concatenate([rng.uniform(-3, 3, (nr, 3)), rng.normal(size=(nr, 3))], axis=1).astype(np.float32)
    v = t[:, :9].reshape(-1, 3, 3); k = rng.integers(0, n, nr // 2); bw = rng.dirichlet([0.3, 0.3, 0.3], nr // 2)
    hr[: nr // 2, 3:] = (v[k] * bw[:, :, None]).sum(1) - hr[: nr // 2, :3]
    src = rng.integers(-1, n, nr).astype(np.int32)
    d_r, d_s = torch.from_numpy(hr).cuda(), torch.from_numpy(src).cuda()
    oi = torch.zeros(nr, dtype=torch.int32, device="cuda"); od = torch.zeros(nr, dtype=torch.float32, device="cuda")
    def hits(fl):
        ctx.closest_hit_device(d_r.data_ptr(), nr, oi.data_ptr(), od.data_ptr(), d_src_idx=d_s.data_ptr(), flags=fl); torch.cuda.synchronize()
        return oi.cpu().numpy().copy(), od.cpu().numpy().view(np.uint32).copy()
    whi, whd = hits(2)
    for var in TWO:
        img, acc = ctx.render(rays, w, h, spp, seed=seed, flags=var | chunks, want_accum=True); st = ctx.stats()
        flat = ctx.render(rays, w, h, 1, mode=capi.MODE_FLAT, flags=var)
        hi, hd = hits(var)
        # opt-in primary-hit reuse (per-pixel closest-hit pre-pass): same bits, w*h*(spp-1) scans fewer
        rimg, racc = ctx.render(rays, w, h, spp, seed=seed, flags=var | chunks | capi.FLAG_PRIMARY_REUSE, want_accum=True); rst = ctx.stats()
        ok = np.array_equal(img, want[0]) and np.array_equal(acc, want[1]) and st["scans_executed"] == ws and np.array_equal(flat, wflat) and np.array_equal(hi, whi) and np.array_equal(hd, whd)
        ok = ok and np.array_equal(rimg, want[0]) and np.array_equal(racc, want[1]) and rst["scans_executed"] == ws - w * h * (spp - 1)
        if not ok:
            fails += 1
            print(f"MISMATCH it {it}: {names[var]} n={n} kind={kind} {w}x{h} spp={spp} seed={seed} chunks={chunks >> 16}: img {np.array_equal(img, want[0])} acc {np.array_equal(acc, want[1])} "
                  f"scans {st['scans_executed']} vs {ws} flat {np.array_equal(flat, wflat)} hits {np.array_equal(hi, whi)} {np.array_equal(hd, whd)}", flush=True)
    if it % 20 == 0: print(f"  {it} scenes, {fails} mismatches", flush=True)
print(f"soak: {it} random scenes x {len(TWO)} two-stage variants (path trace + flat + scan alone) against the exact-only scan: {fails} mismatches", flush=True)
sys.exit(1 if fails else 0)
