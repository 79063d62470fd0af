"""The device arithmetic of the path, function by function, against the oracle (include/spath_hip.h: sphip_selftest_device).
Whole renders already agree bit for bit; these sweeps pin each function on its own, on far more inputs than a render
reaches.  STATED TOLERANCE: 0 (bit patterns compared).

  sincos_glibc     every float in [0, 2*pi] (1.09e9 values: the whole domain geom.h:168-173 can produce) and a sample up to 8
  recip_ieee       every mantissa at 14 exponents incl. the denormal, 2^+-126 and inf/nan ends, both signs, against IEEE division
  philox_uniforms  random and extreme counters/keys
  rand_unit_vec    normals x draws of both generators (24-bit counter uniforms, all 32768 LCG values of frand.h:53-63)
  ray_tri_strict   random, aimed-at-feature, grazing and degenerate (ray, triangle) pairs against geom::ray_intersect
  vec3_rgba        every quantisation boundary
"""
import numpy as np
import pytest

from spath_amd import scene

F = np.float32


def test_oracle_batch_forms_equal_single_calls(O):
    """CPU: spo_device_math_batch is the same code as the single-call functions the KATs pin."""
    import ctypes as C
    L = O.lib()
    x = np.linspace(0, 6.3, 5000, dtype=F)
    sc = O.device_math(0, x, x.size).reshape(-1, 2)
    assert all(sc[i, 0].view(np.uint32) == np.float32(L.spo_sinf(C.c_float(float(x[i])))).view(np.uint32) for i in range(0, 5000, 7))
    assert all(sc[i, 1].view(np.uint32) == np.float32(L.spo_cosf(C.c_float(float(x[i])))).view(np.uint32) for i in range(0, 5000, 7))
    q = np.array([[1, 0, 5, 6, 2], [0xDEADBEEF, 0x12345, 0xFFFFFFFF, 0, 4]], dtype=np.uint32)
    u = O.device_math(2, q, 2).reshape(-1, 2)
    r1, r2 = C.c_double(), C.c_double()
    L.spo_counter_uniforms(C.c_uint64(0x12345DEADBEEF), 0xFFFFFFFF, 0, 4, C.byref(r1), C.byref(r2))
    assert u[1, 0] == r1.value and u[1, 1] == r2.value and 0 <= u.min() and u.max() < 1
    rgba = O.device_math(5, np.array([[0.5, 2.0, -1.0]], dtype=F), 1)
    assert rgba[0] == (128 | (255 << 8))


gpu = pytest.mark.gpu


@gpu
def test_sincos_every_float_in_zero_to_two_pi(hip, O):
    hi = int(np.float32(2.0 * np.pi).view(np.uint32)) + 2          # geom.h:168: float(r * PI * 2.0), r <= 1
    step = 1 << 24
    for lo in range(0, hi, step):
        x = np.arange(lo, min(lo + step, hi), dtype=np.uint32).view(F)
        got, want = hip.selftest(0, x, x.size), O.device_math(0, x, x.size)
        bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
        assert bad.size == 0, (lo, bad[:5], x[bad[:5] // 2])
    x = np.random.default_rng(1).uniform(6.28, 8.0, 1 << 20).astype(F)      # the restated algorithm's fast-reduction range continues to 120
    assert np.array_equal(hip.selftest(0, x, x.size).view(np.uint32), O.device_math(0, x, x.size).view(np.uint32))


@gpu
def test_reciprocal_is_the_ieee_divide(hip):
    man = np.arange(1 << 23, dtype=np.uint32)
    with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
        for e in [0, 1, 2, 3, 64, 100, 126, 127, 128, 150, 200, 251, 252, 253, 254]:          # biased exponents; 0 = denormals
            for sign in (0, 1):
                x = (man | np.uint32(e << 23) | np.uint32(sign << 31)).view(F)
                got, want = hip.selftest(1, x, x.size), (F(1.0) / x).astype(F)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (e, sign)
        sp = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 3.4028235e38, 1.1754944e-38, 1e-14, -1e-14], dtype=F)
        assert np.array_equal(hip.selftest(1, sp, sp.size).view(np.uint32), (F(1.0) / sp).astype(F).view(np.uint32))
        nan = np.array([np.nan], dtype=F)
        assert np.isnan(hip.selftest(1, nan, 1)[0])


@gpu
def test_philox_uniforms(hip, O):
    rng = np.random.default_rng(2)
    q = rng.integers(0, 1 << 32, (1 << 21, 5), dtype=np.uint64).astype(np.uint32)
    q[:64] = np.array([[0, 0, 0, 0, 0], [0xFFFFFFFF] * 5, [1, 0, 0xFFFFFFFF, 0x7FFFFFFE, 4], [0, 1, 2073599, 255, 3]] * 16, dtype=np.uint32)
    got, want = hip.selftest(2, q, q.shape[0]), O.device_math(2, q, q.shape[0])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) and 0.0 <= got.min() and got.max() < 1.0


@gpu
def test_rand_unit_vec(hip, O):
    rng = np.random.default_rng(3)
    n = 1 << 20
    q = np.zeros((n, 5), dtype=np.float64)
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[: n // 8] = np.eye(3)[rng.integers(0, 3, n // 8)] * rng.choice([-1.0, 1.0], (n // 8, 1))       # axis-aligned walls
    q[:, :3] = nrm.astype(F)
    q[:, 3:] = rng.integers(0, 1 << 24, (n, 2)) / 16777216.0                                            # counter uniforms: k / 2^24
    k = np.arange(32768)
    q[:32768, 3] = k / 32767.0; q[:32768, 4] = k[::-1] / 32767.0                                       # every value frand::seed_dist returns
    q[32768:32776, 3:] = [[0, 0], [0, 1], [1, 0], [1, 1], [0.25, 0.5], [0.5, 0.25], [0.75, 16777215 / 16777216], [16777215 / 16777216, 0]]
    got, want = hip.selftest(3, q, n), O.device_math(3, q, n)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _pairs(rng, t, n):
    v = t[:, :9].reshape(-1, 3, 3).astype(np.float64)
    tri = v[rng.integers(0, v.shape[0], n)]
    w = rng.dirichlet([0.25, 0.25, 0.25], n)
    w[: n // 4] = np.eye(3)[rng.integers(0, 3, n // 4)]                   # exactly at vertices
    w[n // 4: n // 2, 2] = 0; w[n // 4: n // 2, :2] = rng.dirichlet([1, 1], n // 4)      # exactly on an edge
    target = (tri * w[:, :, None]).sum(axis=1)
    org = rng.uniform(-3.5, 3.5, (n, 3)) * [1, 0.4, 1] + [0, 0.5, 0]
    d = target - org
    d[: n // 2] /= np.linalg.norm(d[: n // 2], axis=1, keepdims=True)    # the other half stays unnormalised
    miss = rng.random(n) < 0.2
    d[miss] = rng.normal(size=(int(miss.sum()), 3))
    e = tri[:, 1] - tri[:, 0]
    gr = rng.random(n) < 0.05                                              # grazing: origin in the triangle's plane, direction along an edge
    org[gr] = (tri[gr, 2] - 2.0 * e[gr]); d[gr] = e[gr]
    return np.concatenate([org, d, tri.reshape(n, 9)], axis=1).astype(F)


@gpu
def test_ray_tri_strict_is_geom_ray_intersect(hip, O):
    rng = np.random.default_rng(4)
    for t in (scene.closed_room(5000)[0], scene.default_scene()[0], scene.closed_room(600, clutter_scale=10.0)[0]):
        q = _pairs(rng, t, 1 << 20)
        q[:1000, 9:12] = q[:1000, 6:9]                                    # degenerate: v1 == v0
        q[1000:2000, 12:15] = q[1000:2000, 9:12]                          # v2 == v1
        q[2000:2100, 3:6] = 0.0                                           # zero direction
        got, want = hip.selftest(4, q, q.shape[0]), O.device_math(4, q, q.shape[0])
        assert (want > 0).mean() > 0.2
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # magnitudes across the float range (a ~ 1e-14 .. 1e14: both EPSILON tests of geom.h:204,217)
    q = _pairs(rng, scene.closed_room(2000)[0], 1 << 18)
    for s in (1e-9, 1e-5, 1e5, 1e9, 1e15):
        z = q.copy(); z[:, :3] *= F(s); z[:, 6:] *= F(s)
        assert np.array_equal(hip.selftest(4, z, z.shape[0]).view(np.uint32), O.device_math(4, z, z.shape[0]).view(np.uint32)), s


@gpu
def test_vec3_rgba_every_quantisation_boundary(hip, O):
    k = np.arange(0, 256, dtype=np.float64)
    edges = ((k + 0.5) / 255.0).astype(F)                                 # where (x*255 + 0.5) crosses an integer
    near = np.concatenate([np.nextafter(edges, F(-1)), edges, np.nextafter(edges, F(2)), (k / 255.0).astype(F),
                           np.array([-1.0, -0.0, 0.0, 1.0, 1.0000001, 2.0, 1e30, -1e30, 1e-30], dtype=F)])
    rng = np.random.default_rng(5)
    v = np.stack([near, rng.permutation(near), rng.uniform(-0.5, 1.5, near.size).astype(F)], axis=1)
    assert np.array_equal(hip.selftest(5, v, v.shape[0]), O.device_math(5, v, v.shape[0]))
    big = rng.uniform(-0.1, 1.1, (1 << 20, 3)).astype(F)
    assert np.array_equal(hip.selftest(5, big, big.shape[0]), O.device_math(5, big, big.shape[0]))


@gpu
def test_matrix_pipe_side_product_accuracy(hip):
    """sp_cylm_scan.h stage 1: the side product of one (triangle, ray) pair through v_mfma_f32_32x32x16_f16 -- operands split
    into two halves each, three half products per term, 16 products accumulated by the instruction -- against the same 16
    products summed in double.  The conservativeness argument allows 16u of the sum of the |terms| for the
    accumulation (u = 2^-24); measured here: <= 2^-21 (8u), values over the whole range the scaling produces."""
    f16 = np.float16
    rng = np.random.default_rng(6)
    n = 300000
    q = np.zeros((n, 12), dtype=np.float32)
    q[:, :5] = rng.uniform(-16, 16, (n, 5)) * 10.0 ** rng.uniform(-4, 0, (n, 5))            # 16 b, 16 c, 16 Mc'/S
    q[:, 5:10] = rng.uniform(-32, 32, (n, 5)) * 10.0 ** rng.uniform(-4, 0, (n, 5))          # scaled P_b, P_c, -dir
    q[: n // 8, 5:7] *= 512.0                                                                 # rays far from the origin: up to 2^14
    q[:, 10] = f16(rng.uniform(-30, 30, n) * 10.0 ** rng.uniform(-3, 0, n)).astype(np.float32)
    got = hip.selftest(6, q, n).reshape(-1, 2).astype(np.float64)
    # host reference of the same sum (halves by numpy's round-to-nearest-even float16)
    ref = 16.0 * q[:, 10].astype(np.float64)
    for k in range(5):
        th = q[:, k].astype(f16); tl = (q[:, k] - th.astype(np.float32)).astype(f16)
        rh = q[:, 5 + k].astype(f16); rl = (q[:, 5 + k] - rh.astype(np.float32)).astype(f16)
        ref += th.astype(np.float64) * rh.astype(np.float64) + tl.astype(np.float64) * rh.astype(np.float64) + th.astype(np.float64) * rl.astype(np.float64)
    mag = 16.0 * np.abs(q[:, 10].astype(np.float64)) + (np.abs(q[:, :5].astype(np.float64)) * np.abs(q[:, 5:10].astype(np.float64))).sum(1)
    assert (np.abs(got[:, 1] - ref) <= 2.0 ** -23 * mag + 1e-30).all()                        # the device's own double sum = the host's (its float rounding aside)
    err = np.abs(got[:, 0] - ref) / np.maximum(mag, 1e-30)
    assert err.max() <= 2.0 ** -21, (err.max(), np.log2(err.max()))
    # and the splitting itself: the three half products reproduce the float product to 3 x 2^-22 (+ the subnormal floor of the halves)
    prod = 16.0 * q[:, 10].astype(np.float64) + (q[:, :5].astype(np.float64) * q[:, 5:10].astype(np.float64)).sum(1)
    floor = 2.0 ** -24 * (np.abs(q[:, :5]).sum(1) + np.abs(q[:, 5:10]).sum(1))
    assert (np.abs(ref - prod) <= 3.0 * 2.0 ** -22 * mag + floor).all()


@gpu
def test_matrix_pipe_keeps_subnormal_halves(hip):
    """The error budget of sp_cylm_scan.h's stage 1 counts on two things the hardware is free to do otherwise: the float -> half
    conversion must keep subnormal halves (values below 2^-14 become multiples of 2^-24 instead of zero: the `lo` parts of small
    operands), and the f16 matrix instruction must multiply them as such.  One operand pair per item carries a product that exists
    ONLY if both hold: a subnormal half (2^-24 ... 2^-15) times a large half; everything else in the item is zero."""
    n = 4096
    rng = np.random.default_rng(66)
    q = np.zeros((n, 12), dtype=np.float32)
    k = rng.integers(1, 1024, n)                                  # subnormal halves: k * 2^-24, k < 1024
    big = np.float16(rng.uniform(256, 2048, n)).astype(np.float32)
    slot = rng.integers(0, 5, n)
    side = rng.random(n) < 0.5                                    # the subnormal on the triangle side or on the ray side
    sub = (k * 2.0 ** -24).astype(np.float32)
    q[np.arange(n), slot] = np.where(side, sub, big)
    q[np.arange(n), 5 + slot] = np.where(side, big, sub)
    got = hip.selftest(6, q, n).reshape(-1, 2).astype(np.float64)
    want = sub.astype(np.float64) * big.astype(np.float64)        # exact in double, and in the instruction's f32 accumulator
    assert (want > 0).all()
    assert np.array_equal(got[:, 0], want), (np.abs(got[:, 0] - want).max(), int((got[:, 0] == 0).sum()))
    # and a `lo` part that is itself subnormal: v = hi + lo with lo = k * 2^-24 must come through the split intact
    hi = np.float16(rng.uniform(0.01, 0.03, n)).astype(np.float32)            # ulp(hi) = 2^-17 .. 2^-16: lo < 2^-17 is subnormal
    lo = (rng.integers(4, 64, n) * 2.0 ** -24).astype(np.float32)
    q[:] = 0
    q[np.arange(n), slot] = hi + lo                                            # exact in float
    q[np.arange(n), 5 + slot] = big
    got = hip.selftest(6, q, n).reshape(-1, 2).astype(np.float64)
    want = (hi.astype(np.float64) + lo.astype(np.float64)) * big.astype(np.float64)
    # (the two products are added in the f32 accumulator: one rounding; a flushed `lo` would be off by 2^-18 or more)
    assert (np.abs(got[:, 0] - want) <= 2.0 ** -22 * want).all(), (np.abs(got[:, 0] - want) / want).max()
    assert (np.abs(hi.astype(np.float64) * big - want) > 2.0 ** -20 * want).all()              # the check has teeth: dropping lo would show

