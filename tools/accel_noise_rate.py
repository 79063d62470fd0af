"""How often does the opt-in acceleration structure (a geometric cull) disagree with the brute-force scan (the reference's float
evaluation)?  Random rays and surface-bounce rays against closed_room(10000); every disagreement is classified: is the brute-force
hit a NOISE ACCEPT of geom::ray_intersect (the reported hit point lies outside the triangle by far more than rounding)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi, scene
ctx = capi.Context(0)
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
t, m = scene.closed_room(nt)
ctx.set_scene(t, m)
rng = np.random.default_rng(1)
tot = dis = noise = filt_bad = 0
B = 1 << 22
v = t[:, :9].reshape(-1, 3, 3).astype(np.float64)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    d = rng.normal(size=(B, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    if it % 3 == 0:      # anywhere in the room, any direction
        o = rng.uniform(-3.9, 3.9, (B, 3)) * [1, 0.5, 1] + [0, 0.5, 0]
    elif it % 3 == 1:    # starting ON a triangle, like a bounce ray
        k = rng.integers(0, nt, B); w = rng.dirichlet([1, 1, 1], B)
        o = (v[k] * w[:, :, None]).sum(1)
    else:                # the noise regime on purpose: the ray lies in the PLANE of a triangle (to float rounding), passing it at a distance
        k = rng.integers(14, nt, B)
        e1, e2 = v[k, 1] - v[k, 0], v[k, 2] - v[k, 0]
        ab = rng.uniform(-60, 60, (B, 2))                                  # barycentric-ish coordinates far outside [0,1]: up to metres away
        o = v[k, 0] + e1 * ab[:, :1] + e2 * ab[:, 1:]
        cd = rng.normal(size=(B, 2))
        d = e1 * cd[:, :1] + e2 * cd[:, 1:]; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    d_r = torch.from_numpy(rays).cuda()
    out = {}
    for name, fl in (("brute", 2), ("accel", capi.FLAG_ACCEL), ("cyl", 11), ("slab", 3)):
        oi = torch.zeros(B, dtype=torch.int32, device="cuda"); od = torch.zeros(B, dtype=torch.float32, device="cuda")
        ctx.closest_hit_device(d_r.data_ptr(), B, oi.data_ptr(), od.data_ptr(), flags=fl); torch.cuda.synchronize()
        out[name] = (oi.cpu().numpy(), od.cpu().numpy())
    bi, bd = out["brute"]; ai, ad = out["accel"]
    for name in ("cyl", "slab"):          # the two-stage scans of the product path must reproduce the exact scan in this regime too: bit for bit
        fi, fd = out[name]
        nbad = int(((fi != bi) | (fd.view(np.uint32) != bd.view(np.uint32))).sum())
        if nbad: print(f"  !!! two-stage scan {name} differs from the exact scan on {nbad} rays (family {it % 3})", flush=True)
        filt_bad += nbad
    bad = np.flatnonzero((bi != ai) | (bd.view(np.uint32) != ad.view(np.uint32)))
    tot += B; dis += bad.size
    for j in bad:
        # the brute-force hit: where does the reported point lie in the triangle's barycentric frame (double precision)?
        tri = v[bi[j]] if bi[j] >= 0 else None
        if tri is None: continue
        p = rays[j, :3].astype(np.float64) + rays[j, 3:].astype(np.float64) * float(bd[j])
        e1, e2 = tri[1] - tri[0], tri[2] - tri[0]
        n = np.cross(e1, e2); A = np.dot(n, n)
        uu = np.dot(np.cross(p - tri[0], e2), n) / A; vv = np.dot(np.cross(e1, p - tri[0]), n) / A
        off_plane = abs(np.dot(p - tri[0], n)) / np.sqrt(A)
        outside = (uu < -1e-3) or (vv < -1e-3) or (uu + vv > 1 + 1e-3) or off_plane > 1e-3 * max(1.0, np.linalg.norm(p))
        noise += int(outside)
        if dis <= 12:
            print(f"  ray {j}: brute ({bi[j]}, {bd[j]:.6g}) accel ({ai[j]}, {ad[j]:.6g}); brute-force point at barycentric ({uu:.4f}, {vv:.4f}), {off_plane:.3g} off the plane -> {'noise accept' if outside else 'GEOMETRIC HIT MISSED'}", flush=True)
print(f"two-stage scans (cylinder filter rpl_cyl4, slab filter rpl_filter2) vs exact scan: {filt_bad} differing rays of {tot} x 2", flush=True)
print(f"{nt} triangles, {tot} rays: {dis} disagreements ({dis / tot:.3g} of the rays), {noise} of them noise accepts of the reference's float test, {dis - noise} geometric hits missed", flush=True)
