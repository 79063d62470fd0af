"""First contact of the cylinder-filter scan variants (sp_cyl_scan.h) on a GPU: parity against the exact scan and the oracle,
then timings against the first-generation filter variants.  python tools/cyl_check.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from spath_amd import capi, scene, view
from oracle import oracle as O

NEW = [9, 10, 11, 12, 13, 14, 15, 16]
names = {v: k for k, v in capi.kernel_variants().items()}
ctx = capi.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ok = True

def hits(rays, src, flags):
    n = rays.shape[0]
    d_r, d_s = dev(rays), dev(src)
    oi = torch.zeros(n, dtype=torch.int32, device="cuda"); od = torch.zeros(n, dtype=torch.float32, device="cuda")
    ctx.closest_hit_device(d_r.data_ptr(), n, oi.data_ptr(), od.data_ptr(), d_src_idx=d_s.data_ptr(), flags=flags)
    torch.cuda.synchronize()
    return oi.cpu().numpy(), od.cpu().numpy().view(np.uint32), ctx.stats()["kernel_ms"]

# 1. hit for hit against the exact scan: several scenes, random + aimed rays
rng = np.random.default_rng(5)
for name, (t, m) in {"closed10k": scene.closed_room(10000), "bigtris600": scene.closed_room(600, clutter_scale=10.0), "open300": scene.open_clutter(300),
                     "default7x10": tuple(np.concatenate([a] * 10) for a in scene.default_scene()), "odd257": scene.closed_room(257), "n256": scene.closed_room(256),
                     "closed100k": scene.closed_room(100000)}.items():
    n = 200000 if t.shape[0] <= 10000 else 60000
    rays = np.concatenate([rng.uniform(-3, 3, (n, 3)) * [1, 0.4, 1] + [0, 0.5, 0], rng.normal(size=(n, 3))], axis=1).astype(np.float32)
    v = t[:, :9].reshape(-1, 3, 3)
    sel = rng.integers(0, t.shape[0], n // 2)
    w = rng.dirichlet([0.3, 0.3, 0.3], n // 2)                 # near edges and vertices
    target = (v[sel] * w[:, :, None]).sum(axis=1)
    rays[: n // 2, 3:] = (target - rays[: n // 2, :3])
    src = rng.integers(-1, t.shape[0], n).astype(np.int32)
    ctx.set_scene(t, m)
    wi, wd, ms0 = hits(rays, src, 2)
    line = [f"{name:12s} exact {ms0:7.2f} ms"]
    for var in [3] + NEW:
        gi, gd, ms = hits(rays, src, var)
        good = np.array_equal(gi, wi) and np.array_equal(gd, wd)
        ok &= good
        line.append(f"{names[var]} {'ok' if good else 'MISMATCH %d' % (gi != wi).sum()} {ms:6.2f}")
    print(" | ".join(line), flush=True)

# 2. path tracing against the oracle (float accumulators, RGBA8, scan counts)
for name, (t, m), w, h, spp in [("closed1k", scene.closed_room(1000), 96, 64, 5), ("default", scene.default_scene() , 64, 48, 4), ("bigtris", scene.closed_room(600, clutter_scale=10.0), 48, 36, 3)]:
    rays = view.Camera(w, h).get_viewport()
    ctx.set_scene(t, m)
    wimg, wacc, wsc = O.render_counter(rays, t, m, spp, 7)
    wflat = O.render_flat(rays, w, h, t, m)
    for var in NEW:
        img, acc = ctx.render(rays, w, h, spp, seed=7, flags=var, want_accum=True)
        st = ctx.stats()
        flat = ctx.render(rays, w, h, 1, mode=capi.MODE_FLAT, flags=var)
        good = np.array_equal(img, wimg) and np.array_equal(acc, wacc) and st["scans_executed"] == wsc and np.array_equal(flat, wflat)
        ok &= good
        print(f"pt {name:9s} {names[var]:10s} {'bit-exact vs oracle' if good else 'MISMATCH'} (ran variant {st['kernel_variant']}, launches {st['n_launches']})", flush=True)

# 3. speed: the bench frame at reduced spp
if len(sys.argv) < 2:
    t, m = scene.closed_room(10000)
    ctx.set_scene(t, m)
    w, h, spp = 1920, 1080, 16
    rays = view.Camera(w, h).get_viewport()
    d_t, d_m, d_r = dev(t), dev(m), dev(rays)
    ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), t.shape[0], 0)
    out = torch.zeros(w * h, 4, dtype=torch.uint8, device="cuda")
    ref = None
    for var in [6, 13, 15, 14]:
        best = 1e9
        for rep in range(2):
            ctx.render_device(d_r.data_ptr(), w * h, spp, out.data_ptr(), flags=var)
            torch.cuda.synchronize()
            st = ctx.stats(); best = min(best, st["kernel_ms"])
        img = out.cpu().numpy()
        if ref is None: ref = img.copy()
        same = np.array_equal(img, ref); ok &= same
        print(f"1080p x {spp} spp  {names[var]:12s} {best:8.1f} ms  {w*h*spp*5/best/1e3:7.1f} Mray/s  {st['scans_executed']*1e4/best/1e9:.3f} T tests/s  image {'== rpl_filter2s' if same else 'DIFFERS'}", flush=True)
print("ALL OK" if ok else "FAILURES", flush=True)
sys.exit(0 if ok else 1)
