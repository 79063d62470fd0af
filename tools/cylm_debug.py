"""Small-scene hunt for rpl_cylm (variant 16) against the exact-only scan in the scan-alone mode; prints the first mismatching rays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = capi.Context(0)
VAR = int(sys.argv[2]) if len(sys.argv) > 2 else 16
found = 0
for it in range(4000):
    n = int(rng.integers(1, 40))
    kind = int(rng.integers(1, 3))
    if kind == 1 and n >= 7: t, m = scene.open_clutter(n, seed=int(rng.integers(1, 1 << 30)))
    else:
        scale = 10.0 ** rng.uniform(-2.5, 0.5, (n, 1)); ctr = rng.uniform(-2, 2, (n, 3)) * [1, 0.6, 1]
        t = np.zeros((n, 12), dtype=np.float32)
        for k in range(3): t[:, 3 * k:3 * k + 3] = ctr + rng.normal(size=(n, 3)) * scale
        t = scene.flat_normals(t); t[:, 9:12] = np.nan_to_num(t[:, 9:12])
        m = np.zeros((n, 6), dtype=np.float32); m[:, :3] = 0.5
    ctx.set_scene(t, m)
    nr = 20000
    hr = np.concatenate([rng.uniform(-3, 3, (nr, 3)), rng.normal(size=(nr, 3))], axis=1).astype(np.float32)
    v = t[:, :9].reshape(-1, 3, 3); k = rng.integers(0, n, nr // 2); bw = rng.dirichlet([0.3, 0.3, 0.3], nr // 2)
    hr[: nr // 2, 3:] = (v[k] * bw[:, :, None]).sum(1) - hr[: nr // 2, :3]
    src = rng.integers(-1, n, nr).astype(np.int32)
    d_r, d_s = torch.from_numpy(hr).cuda(), torch.from_numpy(src).cuda()
    oi = torch.zeros(nr, dtype=torch.int32, device="cuda"); od = torch.zeros(nr, dtype=torch.float32, device="cuda")
    def hits(fl):
        ctx.closest_hit_device(d_r.data_ptr(), nr, oi.data_ptr(), od.data_ptr(), d_src_idx=d_s.data_ptr(), flags=fl); torch.cuda.synchronize()
        return oi.cpu().numpy().copy(), od.cpu().numpy().copy()
    wi, wd = hits(2); gi, gd = hits(VAR)
    bad = np.nonzero((wi != gi) | (wd.view(np.uint32) != gd.view(np.uint32)))[0]
    if bad.size:
        found += 1
        print(f"it {it} n={n} kind={kind}: {bad.size} rays differ; rv-ish max|v|={np.abs(t[:, :9]).max():.4g}", flush=True)
        for b in bad[:4]:
            print(f"  ray {b}: o={hr[b, :3]} d={hr[b, 3:]} |d|={np.linalg.norm(hr[b, 3:]):.4g} src={src[b]} want ({wi[b]}, {wd[b]:.6g}) got ({gi[b]}, {gd[b]:.6g})", flush=True)
            if wi[b] >= 0: print(f"    tri {wi[b]}: {t[wi[b], :9]}", flush=True)
        b = int(bad[0])
        for cnt in (1, 2, 64, 65, 200):
            rr = np.repeat(hr[b:b + 1], cnt, axis=0).copy(); ss = np.repeat(src[b:b + 1], cnt).copy()
            d_r2, d_s2 = torch.from_numpy(rr).cuda(), torch.from_numpy(ss).cuda()
            oi2 = torch.zeros(cnt, dtype=torch.int32, device="cuda"); od2 = torch.zeros(cnt, dtype=torch.float32, device="cuda")
            ctx.closest_hit_device(d_r2.data_ptr(), cnt, oi2.data_ptr(), od2.data_ptr(), d_src_idx=d_s2.data_ptr(), flags=VAR); torch.cuda.synchronize()
            print(f"    the same ray x{cnt}: idx {np.unique(oi2.cpu().numpy())}", flush=True)
        # neighbours in its wave
        w0 = (b // 64) * 64
        rr = hr[w0:w0 + 64].copy(); ss = src[w0:w0 + 64].copy()
        d_r2, d_s2 = torch.from_numpy(rr).cuda(), torch.from_numpy(ss).cuda()
        oi2 = torch.zeros(64, dtype=torch.int32, device="cuda"); od2 = torch.zeros(64, dtype=torch.float32, device="cuda")
        ctx.closest_hit_device(d_r2.data_ptr(), 64, oi2.data_ptr(), od2.data_ptr(), d_src_idx=d_s2.data_ptr(), flags=VAR); torch.cuda.synchronize()
        print(f"    its wave alone: lane {b - w0} -> {int(oi2[b - w0])}", flush=True)
        np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", f"cylm_case{found}.npz"), t=t, m=m, ray=hr[b], src=src[b], want=wi[b])
        if found >= 3: break
print("done", it, "scenes,", found, "with mismatches", flush=True)
