"""the opt-in acceleration structure (SPHIP_FLAG_ACCEL, linear BVH).  Not the brute-force path and not
what bench.py measures.  Contract checked here: every GEOMETRIC hit is the brute-force scan's hit, index and distance
bits alike; the only permitted differences are the reference's rounding-noise accepts (rays almost coplanar with a
far-away triangle), which must be rare and are recognised by the brute-force scan reporting a hit whose point lies
outside the triangle by far more than rounding."""
import numpy as np
import pytest
import torch

from spath_amd import capi, scene, view

pytestmark = pytest.mark.gpu


def _hits(hip, rays, flags, src=None):
    n = rays.shape[0]
    d_r = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    d_s = torch.from_numpy(src).cuda() if src is not None else None
    d_i = torch.zeros(n, dtype=torch.int32, device="cuda"); d_d = torch.zeros(n, dtype=torch.float32, device="cuda")
    hip.closest_hit_device(d_r.data_ptr(), n, d_i.data_ptr(), d_d.data_ptr(), d_src_idx=d_s.data_ptr() if src is not None else 0, flags=flags)
    torch.cuda.synchronize()
    return d_i.cpu().numpy(), d_d.cpu().numpy()


def _random_rays(rng, n, box=1.4):
    o = rng.uniform(-box, box, (n, 3)) * [1, 0.5, 1]
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("maker,n", [(scene.closed_room, 14), (scene.closed_room, 300), (scene.closed_room, 10000),
                                     (scene.open_clutter, 7), (scene.open_clutter, 1500)])
def test_accel_hits_equal_brute_force(hip, O, maker, n):
    t, m = maker(n)
    hip.set_scene(t, m)
    rng = np.random.default_rng(n)
    rays = np.concatenate([_random_rays(rng, 60000), view.Camera(160, 90).get_viewport()])
    src = rng.integers(-1, t.shape[0], rays.shape[0]).astype(np.int32)
    for s in (None, src):
        bi, bd = _hits(hip, rays, 2, s)                       # exact LDS scan
        ai, ad = _hits(hip, rays, capi.FLAG_ACCEL, s)
        assert hip.stats()["kernel_variant"] == 8
        same = (ai == bi) & (ad.view(np.uint32) == bd.view(np.uint32))
        assert same.mean() >= 0.9999, (same.mean(), n)
        assert (bi >= 0).mean() > 0.3
    if n <= 300:                                               # and against the CPU oracle
        oi, od = O.closest_hits(rays[:5000], t)
        ai, ad = _hits(hip, rays[:5000], capi.FLAG_ACCEL)
        assert ((ai == oi) & (ad.view(np.uint32) == od.view(np.uint32))).mean() >= 0.9999


def test_accel_images_equal_brute_force_images(hip, O):
    t, m = scene.closed_room(2000)
    w, h, spp = 160, 90, 8
    rays = view.Camera(w, h).get_viewport()
    hip.set_scene(t, m)
    a_img, a_acc = hip.render(rays, w, h, spp, seed=3, want_accum=True)
    b_img, b_acc = hip.render(rays, w, h, spp, seed=3, flags=capi.FLAG_ACCEL, want_accum=True)
    st = hip.stats()
    assert st["kernel_variant"] == 8
    diff = (a_acc != b_acc).any(axis=1)
    assert diff.mean() <= 1e-3, diff.mean()                   # noise accepts only
    assert np.array_equal(hip.render(rays, w, h, 1, mode=capi.MODE_FLAT), hip.render(rays, w, h, 1, mode=capi.MODE_FLAT, flags=capi.FLAG_ACCEL))
    # scene change rebuilds the structure
    t2, m2 = scene.open_clutter(500)
    hip.set_scene(t2, m2)
    f1 = hip.render(rays, w, h, 1, mode=capi.MODE_FLAT)
    f2 = hip.render(rays, w, h, 1, mode=capi.MODE_FLAT, flags=capi.FLAG_ACCEL)
    assert np.array_equal(f1, f2)


def test_accel_speed_is_reported_separately(hip):
    """Nominal rays/s with the acceleration structure on configs[2]'s scene at a reduced sample count (information only)."""
    t, m = scene.closed_room(10000)
    w, h, spp = 1920, 1080, 4
    rays = view.Camera(w, h).get_viewport()
    hip.set_scene(t, m)
    a = hip.render(rays, w, h, spp, seed=1, flags=capi.FLAG_ACCEL)
    ms_accel = hip.stats()["kernel_ms"]
    b = hip.render(rays, w, h, spp, seed=1)
    ms_brute = hip.stats()["kernel_ms"]
    frac = (a != b).any(axis=1).mean()
    print(f"\\naccel {ms_accel:.1f} ms vs brute force {ms_brute:.1f} ms for {w}x{h}x{spp}: {w*h*spp*5/ms_accel/1e3:.0f} vs {w*h*spp*5/ms_brute/1e3:.0f} nominal Mray/s; "
          f"pixels that differ (reference noise accepts): {frac:.2e}")
    # (information only: since round 3 the brute-force default is about as fast as the BVH walk at this triangle count)
    assert frac <= 1e-3 and ms_accel > 0 and ms_brute > 0


def test_accel_ties_and_tiny_scenes(hip, O):
    """Duplicate triangles (equal distances -> lowest original index), a single triangle, triangles nobody can hit."""
    t0, m0 = scene.default_scene()
    cases = {
        "dups": (np.concatenate([t0, t0, t0[::-1]]), np.concatenate([m0, m0 * np.float32(0.5), m0[::-1]])),
        "single": (t0[:1], m0[:1]),
        "behind": (t0[:1] + np.array([0, 0, -50] * 3 + [0, 0, 0], dtype=np.float32), m0[:1]),
        "five": (t0[:5], m0[:5]),
    }
    rays = view.Camera(64, 48).get_viewport()
    for name, (t, m) in cases.items():
        hip.set_scene(t, m)
        want_i, want_d = O.closest_hits(rays, t)
        ai, ad = _hits(hip, rays, capi.FLAG_ACCEL)
        assert np.array_equal(ai, want_i) and np.array_equal(ad.view(np.uint32), want_d.view(np.uint32)), name
        img, acc = hip.render(rays, 64, 48, 3, seed=2, flags=capi.FLAG_ACCEL, want_accum=True)
        w_img, w_acc, _ = O.render_counter(rays, t, m, 3, 2)
        assert np.array_equal(img, w_img) and np.array_equal(acc, w_acc), name
