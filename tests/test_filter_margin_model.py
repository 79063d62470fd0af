"""CPU-side check of the two-stage scan's error model, independent of the GPU.

For random and adversarial (ray, triangle) pairs it evaluates, in float32 numpy with the reference's operation
order, the quantities the strict test computes (a_f, sh_f = s.h, vq_f = dir.q) and, with an emulated float32
FMA chain, the filter's gm, t (g0 = gm + t, g1 = gm - t) for each of the three slabs, and checks
  1. the measured discrepancy |g0*|base| - strict value| stays far below the bound D = 36u |dir| (|pos|+|v0|+|v1|)
     the derivation allows (44u for the third slab), and
  2. the filter decision  fl(|gm| - |t|) > Dq  never fires on a pair the strict test accepts.
"""
import numpy as np

F = np.float32
U = 2.0 ** -24


def fma(a, b, c):
    # float32 fused multiply-add emulated through float64 (the product of two floats is exact in double)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)


def dot(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def strict(pos, d, v0, v1, v2):
    """geom::ray_intersect (geom.h:197-222) in float32, returning the accept mask and the intermediates."""
    e1, e2 = v1 - v0, v2 - v0
    h = cross(d, e2)
    a = dot(e1, h)
    with np.errstate(all="ignore"):
        f = (1.0 / a.astype(np.float64)).astype(F)
        s = pos - v0
        sh = dot(s, h)
        u = f * sh
        q = cross(s, e1)
        vq = dot(d, q)
        v = f * vq
        dist = f * dot(e2, q)
    eps = F(1e-14)
    acc = ~((a > -eps) & (a < eps)) & ~((u < 0) | (u > 1)) & ~((v < 0) | ((u + v) > 1)) & (dist > eps) & (dist.astype(np.float64) < 1.0 / float(eps))
    return acc, a, sh, vq, e1, e2


def filter_g(pos, d, w, p0, p1):
    """k_repack_filter + slab_survives arithmetic: mid-line records rounded to float32, gm and t by FMA chains."""
    wn = (w / np.linalg.norm(w, axis=1, keepdims=True)).astype(F)
    wd = wn.astype(np.float64)
    p0d, p1d = p0.astype(np.float64), p1.astype(np.float64)
    Mc = np.cross(wd, 0.5 * (p0d + p1d)).astype(F)
    h = np.cross(wd, 0.5 * (p1d - p0d)).astype(F)
    P = cross(pos, d)
    gm = wn[:, 0] * P[:, 0]
    gm = fma(wn[:, 1], P[:, 1], gm)
    gm = fma(wn[:, 2], P[:, 2], gm)
    gm = fma(-d[:, 0], Mc[:, 0], gm)
    gm = fma(-d[:, 1], Mc[:, 1], gm)
    gm = fma(-d[:, 2], Mc[:, 2], gm)
    t = d[:, 0] * h[:, 0]
    t = fma(d[:, 1], h[:, 1], t)
    t = fma(d[:, 2], h[:, 2], t)
    return gm, t


def make_pairs(rng, n):
    scale = 10.0 ** rng.uniform(-2, 1, (n, 1))
    ctr = rng.normal(size=(n, 3)) * 3
    v = [(ctr + rng.normal(size=(n, 3)) * scale).astype(F) for _ in range(3)]
    bc = rng.dirichlet([0.4, 0.4, 0.4], n)
    tgt = v[0] * bc[:, :1] + v[1] * bc[:, 1:2] + v[2] * bc[:, 2:3]
    tgt = tgt + rng.normal(size=(n, 3)) * scale * 0.2 * (rng.random((n, 1)) < 0.5)
    pos = (rng.normal(size=(n, 3)) * 4).astype(F)
    d = tgt - pos
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
    return pos, d, v[0], v[1], v[2]


def test_error_model_and_conservativeness():
    rng = np.random.default_rng(77)
    n = 400_000
    pos, d, v0, v1, v2 = make_pairs(rng, n)
    acc, a, sh, vq, e1, e2 = strict(pos, d, v0, v1, v2)
    assert acc.mean() > 0.2                                    # plenty of accepted pairs, many on edges/vertices
    nd = np.linalg.norm(d.astype(np.float64), axis=1)
    mag = nd * (np.linalg.norm(pos.astype(np.float64), axis=1) + np.linalg.norm(v0.astype(np.float64), axis=1)
                + np.linalg.norm(v1.astype(np.float64), axis=1) + np.linalg.norm(v2.astype(np.float64), axis=1))
    d1 = np.abs(d).sum(axis=1).astype(np.float64)
    p1n = np.abs(pos).sum(axis=1).astype(np.float64)
    rv = max(np.abs(x).sum(axis=1).max() for x in (v0, v1, v2))
    Dq = 2.0 ** -16 * 1.01 * d1 * (p1n + 2.0 * rv)
    c = (e2.astype(np.float64) - e1.astype(np.float64))
    slabs = {
        "u": (e2, v0, v1, sh.astype(np.float64), (a.astype(np.float64) - sh.astype(np.float64)), 36.0),
        "v": (e1, v0, v2, -vq.astype(np.float64), -(a.astype(np.float64) - vq.astype(np.float64)), 37.0),   # v*a = -(pos-v0).(dir x e1)
        "w": (c.astype(F), v1, v0, None, None, 44.0),
    }
    e1d, e2d = e1.astype(np.float64), e2.astype(np.float64)
    l1, l2, lc = (e1d ** 2).sum(1), (e2d ** 2).sum(1), (c ** 2).sum(1)
    chosen = np.where((l2 >= l1) & (l2 >= lc), "u", np.where(l1 >= lc, "v", "w"))      # k_repack_filter's rule: the longest edge
    seen = 0
    for name, (w, p0, p1, x0, x1, bound_u) in slabs.items():
        gm, t = filter_g(pos, d, w.astype(F), p0, p1)
        g0, g1 = gm.astype(np.float64) + t.astype(np.float64), gm.astype(np.float64) - t.astype(np.float64)
        base = np.linalg.norm(w.astype(np.float64), axis=1)
        if name == "w":
            X = sh.astype(np.float64) + vq.astype(np.float64)          # sh + dir.q = s.(dir x c) = (pos-v0).(dir x c)
            # the filter uses p0 = v1: (pos-v1).(dir x c) = X - a ; p1 = v0: X
            x0, x1 = X - a.astype(np.float64), -X
        ok = np.isfinite(g0) & np.isfinite(g1) & (base > 0) & (chosen == name)       # each slab is only ever used on its own triangles
        seen += int(ok.sum())
        r0 = np.abs(g0[ok] * base[ok] - x0[ok]) / (U * base[ok] * mag[ok])
        r1 = np.abs((-g1[ok]) * base[ok] - x1[ok]) / (U * base[ok] * mag[ok])
        assert r0.max() < bound_u and r1.max() < bound_u, (name, r0.max(), r1.max())
        assert max(r0.max(), r1.max()) < 12.0, (name, r0.max(), r1.max())     # in practice an order of magnitude below the bound
        with np.errstate(all="ignore"):
            reject = (np.abs(gm) - np.abs(t)).astype(np.float64) > Dq          # float32 subtraction, as slab_survives does it
        assert not (reject & acc & ok).any(), (name, int((reject & acc & ok).sum()))
        assert reject[ok].mean() > 0.1                                        # and it does reject
        print(f"slab {name}: {int(ok.sum())} pairs, max discrepancy {max(r0.max(), r1.max()):.2f} u (bound {bound_u:.0f} u), "
              f"rejects {reject[ok].mean():.3f}, accepted by the strict test {acc[ok].mean():.3f}")
    assert seen == n


def test_longest_edge_choice_matches_kernel_rule():
    """k_repack_filter: ties go to e2, then e1 (l2 >= l1 && l2 >= lc; else l1 >= lc)."""
    rng = np.random.default_rng(3)
    v0, v1, v2 = (rng.normal(size=(1000, 3)).astype(F) for _ in range(3))
    e1, e2 = (v1 - v0).astype(np.float64), (v2 - v0).astype(np.float64)
    c = e2 - e1
    l1, l2, lc = (e1 ** 2).sum(1), (e2 ** 2).sum(1), (c ** 2).sum(1)
    pick = np.where((l2 >= l1) & (l2 >= lc), 2, np.where(l1 >= lc, 1, 3))
    longest = np.maximum(np.maximum(l1, l2), lc)
    assert np.array_equal(np.where(pick == 2, l2, np.where(pick == 1, l1, lc)), longest)


def cyl_records(w, p0, p1):
    """cyl_record of sp_cyl_scan.h in numpy: w scaled so that its dominant component is exactly 1, each coefficient computed in
    double and rounded once; H = |h|_2 rounded up.  Returns (class a, beta, gamma, Mc/w_a [n,3], H)."""
    wd = w.astype(np.float64)
    wd = wd / np.linalg.norm(wd, axis=1, keepdims=True)
    aw = np.abs(wd)
    a = np.where((aw[:, 0] >= aw[:, 1]) & (aw[:, 0] >= aw[:, 2]), 0, np.where(aw[:, 1] >= aw[:, 2], 1, 2))
    b, c = (a + 1) % 3, (a + 2) % 3
    idx = np.arange(w.shape[0])
    s = 1.0 / wd[idx, a]
    p0d, p1d = p0.astype(np.float64), p1.astype(np.float64)
    mc = (np.cross(wd, 0.5 * (p0d + p1d)) * s[:, None]).astype(F)
    hv = np.cross(wd, 0.5 * (p1d - p0d)) * s[:, None]
    H = (np.sqrt((hv ** 2).sum(1)) * (1.0 + 2.0 ** -20)).astype(F)
    return a, b, c, (wd[idx, b] * s).astype(F), (wd[idx, c] * s).astype(F), mc, H


def test_cylinder_filter_is_conservative_and_implies_the_slab_reject():
    """Second-generation stage 1 (sp_cyl_scan.h): x = |gm'| - H*D with gm' the axis-normalised 5-term chain.
      1. it never rejects a pair the strict evaluation accepts;
      2. wherever it rejects, the slab quantity |gm| - |t| of the first generation (unit w, evaluated in double) exceeds the
         margin the proof needs (2 x 44u x |dir|(|pos|+|v0|+|v1|+|v2|) covers every slab), i.e. the cylinder's
         rejections are a subset of the proven slab's with room to spare."""
    rng = np.random.default_rng(78)
    n = 400_000
    pos, d, v0, v1, v2 = make_pairs(rng, n)
    # half the rays unnormalised: the filter takes |dir| as it comes
    d = (d * np.where(rng.random((n, 1)) < 0.5, 1.0, 10.0 ** rng.uniform(-3, 3, (n, 1)))).astype(F)
    acc, a_f, sh, vq, e1, e2 = strict(pos, d, v0, v1, v2)
    assert acc.mean() > 0.2
    e1d, e2d = e1.astype(np.float64), e2.astype(np.float64)
    cd = e2d - e1d
    l1, l2, lc = (e1d ** 2).sum(1), (e2d ** 2).sum(1), (cd ** 2).sum(1)
    use_u, use_v = (l2 >= l1) & (l2 >= lc), ~((l2 >= l1) & (l2 >= lc)) & (l1 >= lc)
    w = np.where(use_u[:, None], e2d, np.where(use_v[:, None], e1d, cd))
    p0 = np.where(use_u[:, None] | use_v[:, None], v0, v1)
    p1 = np.where(use_u[:, None], v1, np.where(use_v[:, None], v2, v0))
    a, b, c, beta, gamma, mc, H = cyl_records(w, p0, p1)
    idx = np.arange(n)
    P = cross(pos, d)
    gm = fma(beta, P[idx, b], P[idx, a])
    gm = fma(gamma, P[idx, c], gm)
    gm = fma(-d[:, 0], mc[:, 0], gm)
    gm = fma(-d[:, 1], mc[:, 1], gm)
    gm = fma(-d[:, 2], mc[:, 2], gm)
    D = (np.sqrt(((d * d)[:, 0] + (d * d)[:, 1]) + (d * d)[:, 2]).astype(F) * F(1.0 + 2.0 ** -21)).astype(F)
    x = fma(-H, D, np.abs(gm))
    d1 = np.abs(d).sum(axis=1).astype(np.float64)
    p1n = np.abs(pos).sum(axis=1).astype(np.float64)
    rv = max(np.abs(v).sum(axis=1).max() for v in (v0, v1, v2))
    Dq = np.maximum((F(2.0 ** -16 * 1.01) * (d1 * (p1n + 2.0 * rv)).astype(F) * F(1.0 + 2.0 ** -20)).astype(F), F(1e-37))
    reject = ~((x - Dq) < 0) & np.isfinite(x)                  # the kernel keeps a pair iff the sign bit of x - Dq' is set
    assert not (reject & acc).any(), int((reject & acc).sum())
    assert 0.01 < reject.mean() < 0.9                       # the pairs are aimed at the triangles: few are rejected, but some are
    # the slab quantity of the same triangle, unit w, in double
    wn = w / np.linalg.norm(w, axis=1, keepdims=True)
    posd, dd = pos.astype(np.float64), d.astype(np.float64)
    Pd = np.cross(posd, dd)
    gmu = (wn * Pd).sum(1) - (dd * np.cross(wn, 0.5 * (p0.astype(np.float64) + p1.astype(np.float64)))).sum(1)
    tu = (dd * np.cross(wn, 0.5 * (p1.astype(np.float64) - p0.astype(np.float64)))).sum(1)
    mag = np.linalg.norm(dd, axis=1) * (np.linalg.norm(posd, axis=1) + np.linalg.norm(v0.astype(np.float64), axis=1)
                                         + np.linalg.norm(v1.astype(np.float64), axis=1) + np.linalg.norm(v2.astype(np.float64), axis=1))
    slack = (np.abs(gmu) - np.abs(tu))[reject] / (U * mag[reject])
    assert slack.min() > 2 * 44.0, slack.min()                  # needed: 88u; the margin leaves >= 256u/sqrt(3) ~ 148u even for the scaled w
    print(f"cylinder filter: rejects {reject.mean():.3f}, accepted by the strict test {acc.mean():.3f}; smallest slab slack among its rejections {slack.min():.0f} u (needed 88 u)")


def _half_split(v):
    hi = v.astype(np.float16)
    lo = (v.astype(F) - hi.astype(F)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def test_matrix_pipe_filter_is_conservative_and_implies_the_slab_reject():
    """Third-generation stage 1 (sp_cylm_scan.h): the same x = |gm'| - H*D with gm' evaluated as the matrix instruction does --
    every operand split into two halves (v = hi + lo), three half products per term, P_a as one half after scaling the ray by
    t = half(P_a)/P_a, everything scaled by exact powers of two (S for lengths, s_d for the direction), accumulated in f32
    (modelled: the 16 products summed in double, then rounded once; the device figure, 2^-21.5 of the sum of the |terms|,
    is measured by tests/test_hip_device_math.py).  Checks as for the cylinder filter: never rejects what the strict
    evaluation accepts; every rejection has the slab slack the proof needs."""
    rng = np.random.default_rng(79)
    n = 400_000
    pos, d, v0, v1, v2 = make_pairs(rng, n)
    d = (d * np.where(rng.random((n, 1)) < 0.5, 1.0, 10.0 ** rng.uniform(-3, 3, (n, 1)))).astype(F)
    acc, a_f, sh, vq, e1, e2 = strict(pos, d, v0, v1, v2)
    e1d, e2d = e1.astype(np.float64), e2.astype(np.float64)
    cd = e2d - e1d
    l1, l2, lc = (e1d ** 2).sum(1), (e2d ** 2).sum(1), (cd ** 2).sum(1)
    use_u, use_v = (l2 >= l1) & (l2 >= lc), ~((l2 >= l1) & (l2 >= lc)) & (l1 >= lc)
    w = np.where(use_u[:, None], e2d, np.where(use_v[:, None], e1d, cd))
    p0 = np.where(use_u[:, None] | use_v[:, None], v0, v1)
    p1 = np.where(use_u[:, None], v1, np.where(use_v[:, None], v2, v0))
    a, b, c, beta, gamma, mc, H = cyl_records(w, p0, p1)
    idx = np.arange(n)
    P = cross(pos, d)
    rv = F(max(np.abs(v).sum(axis=1).max() for v in (v0, v1, v2)))
    S = F(2.0 ** (np.floor(np.log2(float(rv))) + 2))                 # the power of two in (2 Rv, 4 Rv]
    assert 2 * rv <= S <= 4 * rv * (1 + 1e-6)
    dmax = np.abs(d).max(axis=1)
    s_d = (2.0 ** -np.floor(np.log2(dmax.astype(np.float64)))).astype(F)     # dmax * s_d in [1, 2)
    kP, kN, kC = (s_d * F(16.0 / S)).astype(F), (F(16.0) * s_d).astype(F), (s_d * F(256.0 / S)).astype(F)
    D = (np.sqrt(((d * d)[:, 0] + (d * d)[:, 1]) + (d * d)[:, 2]).astype(F) * F(1.0 + 2.0 ** -21)).astype(F)
    d1 = np.abs(d).sum(axis=1).astype(np.float64)
    p1n = np.abs(pos).sum(axis=1).astype(np.float64)
    Dq = np.maximum((F(2.0 ** -16 * 1.01) * (d1 * (p1n + 2.0 * float(rv))).astype(F) * F(1.0 + 2.0 ** -20)).astype(F), F(1e-37))
    Pa_, Pb_, Pc_ = (P[idx, a] * kP).astype(F), (P[idx, b] * kP).astype(F), (P[idx, c] * kP).astype(F)
    ah = Pa_.astype(np.float16).astype(F)
    big = np.abs(Pa_) >= F(2.0 ** -10)
    with np.errstate(all="ignore"):
        t = np.where(big, (ah.astype(np.float64) / Pa_.astype(np.float64)).astype(F), F(1.0)).astype(F)
    extra = np.where(big, F(0.0), (np.abs(Pa_ - ah) * F(16.0)).astype(F)).astype(F)
    ray_vals = [(Pb_ * t).astype(F), (Pc_ * t).astype(F)] + [((-d[:, k]) * kN * t).astype(F) for k in range(3)]
    tri_vals = [(F(16.0) * beta).astype(F), (F(16.0) * gamma).astype(F)] + [(mc[:, k] * F(16.0 / S)).astype(F) for k in range(3)]
    g = 16.0 * ah.astype(np.float64)
    for rvl, tvl in zip(ray_vals, tri_vals):
        rh, rl = _half_split(rvl)
        th, tl = _half_split(tvl)
        g = g + th * rh + tl * rh + th * rl
    g = g.astype(F)
    Hh = (H * F(256.0 / S)).astype(F)
    Dt = (D * s_d * t).astype(F)
    Dqt = (((Dq * kC) * t + extra) * F(1.0 + 2.0 ** -20)).astype(F)
    x = fma(-Hh, Dt, np.abs(g))
    in_range = (p1n + 2.0 * float(rv)) <= 512.0 * float(S)
    reject = ~((x - Dqt) < 0) & np.isfinite(x) & in_range
    assert not (reject & acc).any(), int((reject & acc).sum())
    assert 0.01 < reject.mean() < 0.9
    # the f32 cylinder filter rejects (nearly) the same pairs: the halves cost no selectivity
    gm = fma(beta, P[idx, b], P[idx, a]); gm = fma(gamma, P[idx, c], gm)
    for k in range(3): gm = fma(-d[:, k], mc[:, k], gm)
    rej32 = ~((fma(-H, D, np.abs(gm)) - Dq) < 0)
    assert abs(int(reject.sum()) - int((rej32 & in_range).sum())) <= 1e-3 * n
    wn = w / np.linalg.norm(w, axis=1, keepdims=True)
    posd, dd = pos.astype(np.float64), d.astype(np.float64)
    Pd = np.cross(posd, dd)
    gmu = (wn * Pd).sum(1) - (dd * np.cross(wn, 0.5 * (p0.astype(np.float64) + p1.astype(np.float64)))).sum(1)
    tu = (dd * np.cross(wn, 0.5 * (p1.astype(np.float64) - p0.astype(np.float64)))).sum(1)
    mag = np.linalg.norm(dd, axis=1) * (np.linalg.norm(posd, axis=1) + np.linalg.norm(v0.astype(np.float64), axis=1)
                                         + np.linalg.norm(v1.astype(np.float64), axis=1) + np.linalg.norm(v2.astype(np.float64), axis=1))
    slack = (np.abs(gmu) - np.abs(tu))[reject] / (U * mag[reject])
    assert slack.min() > 2 * 44.0, slack.min()
    print(f"matrix-pipe filter: rejects {reject.mean():.3f} (f32 cylinder filter: {rej32.mean():.3f}); smallest slab slack among its rejections {slack.min():.0f} u (needed 88 u)")


def _matrix_pipe_x(pos, d, v0, v1, v2):
    """The matrix-pipe side product g^ and the scaled H^, D^, Dq^ of every pair, exactly as in the test above (class = dominant
    axis of the longest edge, powers-of-two scalings, t = half(P_a)/P_a, three half products per term)."""
    n = pos.shape[0]
    acc, a_f, sh, vq, e1, e2 = strict(pos, d, v0, v1, v2)
    e1d, e2d = e1.astype(np.float64), e2.astype(np.float64)
    cd = e2d - e1d
    l1, l2, lc = (e1d ** 2).sum(1), (e2d ** 2).sum(1), (cd ** 2).sum(1)
    use_u, use_v = (l2 >= l1) & (l2 >= lc), ~((l2 >= l1) & (l2 >= lc)) & (l1 >= lc)
    w = np.where(use_u[:, None], e2d, np.where(use_v[:, None], e1d, cd))
    p0 = np.where(use_u[:, None] | use_v[:, None], v0, v1)
    p1 = np.where(use_u[:, None], v1, np.where(use_v[:, None], v2, v0))
    a, b, c, beta, gamma, mc, H = cyl_records(w, p0, p1)
    idx = np.arange(n)
    P = cross(pos, d)
    rv = F(max(np.abs(v).sum(axis=1).max() for v in (v0, v1, v2)))
    S = F(2.0 ** (np.floor(np.log2(float(rv))) + 2))
    dmax = np.abs(d).max(axis=1)
    s_d = (2.0 ** -np.floor(np.log2(dmax.astype(np.float64)))).astype(F)
    kP, kN, kC = (s_d * F(16.0 / S)).astype(F), (F(16.0) * s_d).astype(F), (s_d * F(256.0 / S)).astype(F)
    D = (np.sqrt(((d * d)[:, 0] + (d * d)[:, 1]) + (d * d)[:, 2]).astype(F) * F(1.0 + 2.0 ** -21)).astype(F)
    d1 = np.abs(d).sum(axis=1).astype(np.float64)
    p1n = np.abs(pos).sum(axis=1).astype(np.float64)
    Dq = np.maximum((F(2.0 ** -16 * 1.01) * (d1 * (p1n + 2.0 * float(rv))).astype(F) * F(1.0 + 2.0 ** -20)).astype(F), F(1e-37))
    Pa_, Pb_, Pc_ = (P[idx, a] * kP).astype(F), (P[idx, b] * kP).astype(F), (P[idx, c] * kP).astype(F)
    ah = Pa_.astype(np.float16).astype(F)
    big = np.abs(Pa_) >= F(2.0 ** -10)
    with np.errstate(all="ignore"):
        t = np.where(big, (ah.astype(np.float64) / Pa_.astype(np.float64)).astype(F), F(1.0)).astype(F)
    extra = np.where(big, F(0.0), (np.abs(Pa_ - ah) * F(16.0)).astype(F)).astype(F)
    ray_vals = [(Pb_ * t).astype(F), (Pc_ * t).astype(F)] + [((-d[:, k]) * kN * t).astype(F) for k in range(3)]
    tri_vals = [(F(16.0) * beta).astype(F), (F(16.0) * gamma).astype(F)] + [(mc[:, k] * F(16.0 / S)).astype(F) for k in range(3)]
    g = 16.0 * ah.astype(np.float64)
    for rvl, tvl in zip(ray_vals, tri_vals):
        rh, rl = _half_split(rvl)
        th, tl = _half_split(tvl)
        g = g + th * rh + tl * rh + th * rl
    in_range = (p1n + 2.0 * float(rv)) <= 512.0 * float(S)
    return dict(acc=acc, cls=a, g=g.astype(F), Hh=(H * F(256.0 / S)).astype(F), Dt=(D * s_d * t).astype(F),
                Dqt=(((Dq * kC) * t + extra) * F(1.0 + 2.0 ** -20)).astype(F), in_range=in_range)


def _groups_of_four(seed, ng):
    """ng groups of four triangles of (mostly) one class with similar H, each group against one ray -- aimed at its first triangle or
    borrowed from another group -- as _matrix_pipe_x sees them: (q, r) with r() reshaping a per-pair array to [ng, 4]."""
    rng = np.random.default_rng(seed)
    pos, d, v0, v1, v2 = make_pairs(rng, ng)
    d = (d * np.where(rng.random((ng, 1)) < 0.5, 1.0, 10.0 ** rng.uniform(-3, 3, (ng, 1)))).astype(F)
    # half of the rays are borrowed from another group: they pass the four triangles at a distance (what stage 1 is there to reject)
    swap = rng.random(ng) < 0.5
    pos = np.where(swap[:, None], np.roll(pos, 1, axis=0), pos).astype(F)
    d = np.where(swap[:, None], np.roll(d, 1, axis=0), d).astype(F)
    # three more triangles per ray: the first one moved and scaled a little (same longest edge direction: same class, similar H)
    P4, D4, V0, V1, V2 = [pos], [d], [v0], [v1], [v2]
    for k in range(3):
        shift = (rng.normal(size=(ng, 3)) * np.abs(v1 - v0).max(axis=1, keepdims=True) * rng.choice([0.3, 3.0, 30.0], (ng, 1))).astype(F)
        sc = rng.uniform(0.8, 1.25, (ng, 1)).astype(F)
        P4.append(pos); D4.append(d)
        V0.append((v0 + shift).astype(F)); V1.append((v0 + shift + (v1 - v0) * sc).astype(F)); V2.append((v0 + shift + (v2 - v0) * sc).astype(F))
    cat = lambda xs: np.stack(xs, axis=1).reshape(ng * 4, 3)
    return _matrix_pipe_x(cat(P4), cat(D4), cat(V0), cat(V1), cat(V2)), (lambda x: x.reshape(ng, 4))


def test_margin_folded_into_the_bound_rejects_no_more_than_the_subtracted_form():
    """The shipped stage 1 (sp_cylm_scan.h, cylm_tile_bound / cylm_bits / k_cylm_hmax): the margin Dq^ is folded into the cylinder bound,
    x' = fma(-Hmax^, D', min_i |g^_i|) with D' = fl(fl(fma(Dq^, kappa, D^)) (1 + 2^-22)) and kappa = fl(1 / Hmin)(1 + 2^-20), Hmin the
    smallest Hmax^ of the TILE after the small ones were raised to 2^-10 of the largest; reject iff !(x' < 0) -- four instructions
    per group.  Because Hmax^ kappa >= 1, Hmax^ D' >= Hmax^ D^ + Dq^, and x' is rounded once (its sign is exact):
      1. a rejected group contains no pair the strict evaluation accepts;
      2. whatever this form rejects, the form with the margin subtracted (previous test; the one DESIGN.md 4.2/4.3 proves) rejects too;
      3. in tiles of an H-sorted stream it keeps hardly more."""
    ng, G = 96_000, 64                                                   # 64 groups = one 256-triangle tile
    q, r = _groups_of_four(81, ng)
    same = (r(q["cls"]) == r(q["cls"])[:, :1]).all(axis=1)
    m = np.abs(r(q["g"])).min(axis=1)
    Hmax = r(q["Hh"]).max(axis=1)
    Dt, Dqt = r(q["Dt"])[:, 0], r(q["Dqt"])[:, 0]
    ok = np.isfinite(Hmax) & r(q["in_range"])[:, 0] & same
    rej_old = ~((fma(-Hmax, Dt, m) - Dqt) < 0) & ok
    acc = r(q["acc"])
    for order, worst_lost in ((np.argsort(Hmax, kind="stable"), 0.02), (np.arange(ng), 0.6)):      # the prepass's H-sorted stream; an unsorted one
        Hm = Hmax[order].reshape(-1, G).copy()
        top = np.where(np.isfinite(Hm), Hm, F(0)).max(axis=1, keepdims=True)
        floor_ = np.maximum((top * F(2.0 ** -10)).astype(F), F(2.0 ** -60))
        Hm = np.where(np.isfinite(Hm), np.maximum(Hm, floor_), Hm).astype(F)
        hmin = np.where(np.isfinite(Hm), Hm, F(np.inf)).min(axis=1, keepdims=True)
        kappa = ((F(1.0) / hmin).astype(F) * F(1.0 + 2.0 ** -20)).astype(F)
        assert (Hm.astype(np.float64) * kappa.astype(np.float64) >= 1.0)[np.isfinite(Hm)].all()
        Dn, Dq, mm = Dt[order].reshape(-1, G), Dqt[order].reshape(-1, G), m[order].reshape(-1, G)
        Dp = (fma(Dq, np.broadcast_to(kappa, Dq.shape).astype(F), Dn) * F(1.0 + 2.0 ** -22)).astype(F)
        assert (Dp.astype(np.float64) >= Dn.astype(np.float64) + Dq.astype(np.float64) * kappa.astype(np.float64))[np.isfinite(Dp)].all()
        xs = fma(-Hm, Dp, mm)
        rej_new = np.zeros(ng, dtype=bool)
        rej_new[order] = (~(xs < 0) & np.isfinite(xs)).reshape(-1)
        rej_new &= ok
        assert not (rej_new[:, None] & acc).any(), int((rej_new[:, None] & acc).sum())        # 1.
        assert not (rej_new & ~rej_old).any(), int((rej_new & ~rej_old).sum())                # 2.
        lost = 1.0 - rej_new.sum() / max(rej_old.sum(), 1)
        assert lost < worst_lost, lost                                                          # 3.
        print(f"margin folded into the bound: rejects {rej_new.mean():.4f} of the groups, the subtracted form {rej_old.mean():.4f} ({lost:.3%} kept in addition)")


def test_grouped_bound_of_the_matrix_pipe_filter_is_conservative():
    """Round 3's grouped bound (sp_cylm_scan.h): ONE bound per group of four triangles of a class, x = fma(-Hmax, D^, min_i |g^_i|),
    reject the group iff x > Dq^ (the shipped kernel folds Dq^ into D^: test above).  Groups of four triangles of one class tested
    against one ray (aimed at one of the four):
      1. a rejected group contains no pair the strict evaluation accepts;
      2. a rejected group is rejected triangle by triangle too by the per-pair form the proof is written
         for -- fl(m - Hmax D) <= fl(|g_i| - H_i D) because m <= |g_i|, Hmax >= H_i, D > 0 and rounding is monotone -- so the
         grouped bound can only KEEP more;
      3. with the group's H values close together (the prepass sorts every class by H) it keeps hardly more."""
    ng = 100_000
    q, r = _groups_of_four(80, ng)
    same = (r(q["cls"]) == r(q["cls"])[:, :1]).all(axis=1)           # groups of one class (the stream is class-sorted)
    assert same.mean() > 0.5
    # the scaled ray values depend on the ray and the class only
    assert (r(q["Dt"])[same] == r(q["Dt"])[same][:, :1]).all() and (r(q["Dqt"])[same] == r(q["Dqt"])[same][:, :1]).all()
    m = np.abs(r(q["g"])).min(axis=1)
    Hmax = r(q["Hh"]).max(axis=1)
    Dt, Dqt = r(q["Dt"])[:, 0], r(q["Dqt"])[:, 0]
    xg = fma(-Hmax, Dt, m)
    rej_g = ~((xg - Dqt) < 0) & np.isfinite(xg) & r(q["in_range"])[:, 0] & same
    xi = fma(-q["Hh"], q["Dt"], np.abs(q["g"]))
    rej_i = r(~((xi - q["Dqt"]) < 0) & np.isfinite(xi) & q["in_range"])
    acc = r(q["acc"])
    assert acc[:, 0].mean() > 0.1 and acc[:, 1:].mean() > 0.001       # the aimed-at triangle and now and then a neighbour
    assert not (rej_g[:, None] & acc).any(), int((rej_g[:, None] & acc).sum())
    assert not (rej_g[:, None] & ~rej_i).any()                        # 2.
    all_i = rej_i.all(axis=1) & same
    assert 0.02 < rej_g.mean() < 0.9
    lost = 1.0 - rej_g.sum() / max(all_i.sum(), 1)
    assert lost < 0.08, lost                                           # 3. (H within a group spread over a factor 1.56 here; sorted streams are tighter)
    print(f"grouped bound: rejects {rej_g.mean():.3f} of the groups, the four per-pair tests together {all_i.mean():.3f} ({lost:.2%} kept in addition)")

