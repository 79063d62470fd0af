"""Per-pair audit of stage 1 of the default scan ON THE DEVICE (sphip_selftest_stage1).

Every other GPU test observes the conservative filter through the CLOSEST hit: a wrongly rejected pair that is not the nearest
hit of its ray is invisible there.  Here stage 1 alone is run -- the same ray setup, the same fragment code, the same tiles the
render kernels use -- and for every (ray, triangle) pair of the batch the oracle's geom::ray_intersect (reference src/geom.h:197-222)
is evaluated: NO PAIR THE REFERENCE ACCEPTS (d > 0) MAY HAVE ITS BIT CLEAR, neither the group bit the kernels use nor the stronger
per-triangle form of the test (the triangle's own cylinder radius instead of its group's largest).  More than 10^7 pairs per run:
rays aimed at vertices, edge points and interiors, rays in the planes of triangles (the reference's noise accepts), millimetre
triangles seen from metres away (the regime of round 2's float->half conversion defect), scenes scaled by 1e-4 ... 1e20.

A second test runs the same audit on a build with that defect switched back on (-DSP_CYLM_UNPINNED) and expects it to FAIL
there: the negative control that keeps the audit honest.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from spath_amd import capi, scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def _aimed_rays(rng, t, n, origin_box, unnormalised=0.5):
    """n rays: 3/4 aimed at vertices / edge points / interior points of random triangles, 1/4 random directions."""
    v = t[:, :9].reshape(-1, 3, 3).astype(np.float64)
    k = rng.integers(0, v.shape[0], n)
    w = rng.dirichlet([0.3, 0.3, 0.3], n)
    w[: n // 4] = np.eye(3)[rng.integers(0, 3, n // 4)]                                   # exactly at vertices
    w[n // 4: n // 2, 2] = 0; w[n // 4: n // 2, :2] = rng.dirichlet([1, 1], n // 4)       # exactly on an edge
    tgt = (v[k] * w[:, :, None]).sum(axis=1)
    o = rng.uniform(-1, 1, (n, 3)) * origin_box
    d = tgt - o
    rnd = rng.random(n) < 0.25
    d[rnd] = rng.normal(size=(int(rnd.sum()), 3)) * np.linalg.norm(d[rnd], axis=1, keepdims=True)
    nrm = rng.random(n) >= unnormalised
    d[nrm] /= np.maximum(np.linalg.norm(d[nrm], axis=1, keepdims=True), 1e-300)
    return np.concatenate([o, d], axis=1).astype(F)


def _in_plane_rays(rng, t, n):
    """Rays lying (to rounding) in the plane of a triangle and passing it at a distance: where geom::ray_intersect noise-accepts."""
    v = t[:, :9].reshape(-1, 3, 3).astype(np.float64)
    k = rng.integers(0, v.shape[0], n)
    e1, e2 = v[k, 1] - v[k, 0], v[k, 2] - v[k, 0]
    ab = rng.uniform(-40, 40, (n, 2))
    o = v[k, 0] + e1 * ab[:, :1] + e2 * ab[:, 1:]
    cd = rng.normal(size=(n, 2))
    d = e1 * cd[:, :1] + e2 * cd[:, 1:]
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-300)
    return np.concatenate([o, d], axis=1).astype(F)


def _small_far_scene(rng, n):
    """Triangles of 3 mm ... 3 m scattered over a few metres (flat_normals only fills n; the audit never reads it)."""
    scale = 10.0 ** rng.uniform(-2.5, 0.5, (n, 1)); ctr = rng.uniform(-2, 2, (n, 3)) * [1, 0.6, 1]
    t = np.zeros((n, 12), dtype=F)
    for k in range(3):
        t[:, 3 * k:3 * k + 3] = ctr + rng.normal(size=(n, 3)) * scale
    with np.errstate(all="ignore"):
        t = scene.flat_normals(t)
    t[:, 9:12] = np.nan_to_num(t[:, 9:12])
    return t


def audit(hip, O, t, rays, tag, chunk=192):
    """Returns (pairs, accepted, group bits set, triangle bits set); asserts the conservativeness of both forms."""
    n = t.shape[0]
    hip.set_scene(t, np.full((n, 6), 0.5, dtype=F))
    grp, tri, order = hip.selftest_stage1(rays)
    valid = (order >= 0) & (order < n)
    pos_of = np.full(n, -1, dtype=np.int64)
    pos_of[order[valid]] = np.flatnonzero(valid)
    # a triangle without a place in the stream is in the "big" class (at most 64 per scene): no filter, every ray runs the
    # reference's test on it
    big = pos_of < 0
    assert big.sum() <= 64, int(big.sum())
    assert np.array_equal(np.sort(order[valid]), np.flatnonzero(~big)), "every other triangle exactly once"
    # (at the places of the stream that hold a triangle: on padding, H = -inf, a ray whose filter is off compares inf - inf)
    assert not (tri & ~grp)[:, valid].any(), (tag, "a surviving triangle whose group does not survive")
    grp_t, tri_t = grp[:, np.maximum(pos_of, 0)], tri[:, np.maximum(pos_of, 0)]  # [ray, triangle index]
    grp_t[:, big] = True; tri_t[:, big] = True
    accepted = 0
    tv = t[:, :9]
    for r0 in range(0, rays.shape[0], chunk):
        rr = rays[r0:r0 + chunk]
        q = np.concatenate([np.repeat(rr, n, axis=0), np.tile(tv, (rr.shape[0], 1))], axis=1)
        with np.errstate(all="ignore"):
            d = O.device_math(4, q, q.shape[0]).reshape(rr.shape[0], n)
        acc = d > 0.0                                                             # geom::ray_intersect accepted the pair
        accepted += int(acc.sum())
        bad_t = acc & ~tri_t[r0:r0 + chunk]
        bad_g = acc & ~grp_t[r0:r0 + chunk]
        assert not bad_t.any(), (tag, "per-triangle test rejected an accepted pair", np.argwhere(bad_t)[:4] + [r0, 0], int(bad_t.sum()))
        assert not bad_g.any(), (tag, "group test rejected an accepted pair", np.argwhere(bad_g)[:4] + [r0, 0], int(bad_g.sum()))
    return rays.shape[0] * n, accepted, int(grp_t.sum()), int(tri_t.sum())


def _fuzz_soups(which):
    """The triangle soups and aimed rays of tests/test_hip_robustness.py::test_fuzz_filter_scan_against_exact_scan (same generator,
    same seed): soup number 6 is where the build with the conversion defect returns a wrong closest hit."""
    rng = np.random.default_rng(2026)
    for it in range(max(which) + 1):
        n = int(rng.integers(65, 3000))
        size = 10.0 ** rng.uniform(-3, 1.5)
        spread = 10.0 ** rng.uniform(-1, 2)
        ctr = rng.normal(size=(n, 1, 3)) * spread
        v = ctr + rng.normal(size=(n, 3, 3)) * size * (10.0 ** rng.uniform(-2, 0, size=(n, 1, 1)))
        if it % 3 == 0:
            v[:, 2] = v[:, 0] + (v[:, 1] - v[:, 0]) * rng.uniform(0, 1, (n, 1)) + rng.normal(size=(n, 3)) * size * 1e-4
        t = np.zeros((n, 12), dtype=F)
        t[:, :9] = v.reshape(n, 9)
        with np.errstate(all="ignore"):
            t = scene.flat_normals(t)
        nr = 40000
        o = rng.normal(size=(nr, 3)) * spread * 1.5
        bc = rng.dirichlet([0.5, 0.5, 0.5], nr)
        tri = v[rng.integers(0, n, nr)]
        tgt = (tri * bc[:, :, None]).sum(axis=1) + rng.normal(size=(nr, 3)) * size * 0.3 * (rng.random((nr, 1)) < 0.5)
        d = tgt - o
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
        rng.integers(-1, n, nr)                                   # (the fuzz test draws idx_source here)
        if it in which:
            yield it, t, np.concatenate([o, d], axis=1).astype(F)


def run_cases(hip, O, log=None, fuzz_soups=(), fuzz_rays=4096):
    rng = np.random.default_rng(20261005)
    tot_pairs = tot_acc = 0
    cases = []
    for it, t, rays in _fuzz_soups(fuzz_soups):
        cases.append((f"fuzz soup #{it} ({t.shape[0]})", t, rays[:fuzz_rays]))
    t = scene.closed_room(2560)[0]
    cases.append(("closed room, aimed", t, _aimed_rays(rng, t, 1536, [3.5, 1.8, 3.5])))
    cases.append(("closed room, in-plane", t, _in_plane_rays(rng, t, 768)))
    t = scene.closed_room(1500, clutter_scale=10.0)[0]
    cases.append(("large triangles, aimed", t, _aimed_rays(rng, t, 1024, [3.5, 1.8, 3.5])))
    t = scene.open_clutter(700)[0]
    cases.append(("open clutter, aimed + in-plane", t, np.concatenate([_aimed_rays(rng, t, 768, [2, 1, 2]), _in_plane_rays(rng, t, 256)])))
    for it in range(12):                                          # millimetre triangles from metres away, few triangles per scene
        n = int(rng.integers(1, 60)) if it < 8 else int(rng.integers(100, 400))
        t = _small_far_scene(rng, n)
        cases.append((f"small far #{it} ({n})", t, _aimed_rays(rng, t, 4096 if n < 60 else 1024, [3, 3, 3], unnormalised=0.7)))
    base = scene.closed_room(600)[0]
    for s in (1e-4, 1e-2, 1e3, 1e8, 1e20):                        # scales; beyond 1e18 in direction length the filter switches itself off
        t = base.copy(); t[:, :9] *= F(s)
        r = _aimed_rays(rng, t, 512, np.array([3.5, 1.8, 3.5]) * s)
        cases.append((f"scale {s:g}", t, r))
    t = base.copy()
    r = _aimed_rays(rng, t, 512, [3.5, 1.8, 3.5]); r[:, 3:] *= F(1e-12)          # tiny directions
    cases.append(("directions x 1e-12", t, r))
    r = _aimed_rays(rng, t, 512, [2000.0, 900.0, 2000.0])                         # cameras hundreds of radii away (filter off beyond 512 S)
    cases.append(("far origins", t, r))
    for tag, t, rays in cases:
        with np.errstate(all="ignore"):
            pairs, acc, gb, tb = audit(hip, O, t, rays, tag)
        tot_pairs += pairs; tot_acc += acc
        if log is not None:
            log.append(f"{tag}: {pairs} pairs, {acc} accepted by geom::ray_intersect, {gb / pairs:.4%} group bits, {tb / pairs:.4%} triangle bits set")
    return tot_pairs, tot_acc


def test_stage1_never_rejects_a_pair_the_reference_accepts(hip, O):
    log = []
    pairs, acc = run_cases(hip, O, log, fuzz_soups=(1, 6), fuzz_rays=2048)
    print("\n".join(log))
    assert pairs >= 10_000_000, pairs
    assert acc >= 20_000, acc                                     # the batch does contain hits: aimed rays


def _unpinned_lib():
    """The default library with the one-conversion-per-value guard of sp_cylm_scan.h (to_half) compiled out."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    return g.build_unpinned()


def test_audit_fails_on_the_build_with_the_conversion_defect():
    """Negative control: with float->half conversions unpinned (fragment value and remainder from two different conversion
    instructions) the half products are off by an ulp of the half now and then; the audit must catch that.  Runs in a child
    process (another libspath_hip.so cannot be loaded next to the first one under the same ctypes binding)."""
    lib = _unpinned_lib()
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from spath_amd import capi; capi.LIB_PATH = %r\n"
            "from oracle import oracle as O\n"
            "import test_hip_stage1_audit as A\n"
            "hip = capi.Context(0)\n"
            "try:\n"
            "    A.run_cases(hip, O, fuzz_soups=(6,), fuzz_rays=40000)\n"
            "except AssertionError as e:\n"
            "    print('AUDIT-FAILED', str(e)[:300]); sys.exit(3)\n"
            "print('AUDIT-PASSED'); sys.exit(0)\n") % (ROOT, os.path.join(ROOT, "tests"), lib)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert p.returncode == 3 and "AUDIT-FAILED" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-800:])
    assert "rejected an accepted pair" in p.stdout, p.stdout[-500:]
