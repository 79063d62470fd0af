"""Header-level known answers: the C oracle against values dumped from the reference's own inline
functions (tests/golden/golden.json 'kat', produced by oracle/_ref/spath_ref kat; the seed_dist(0),
rand_unit_vec(seed 7), flat_normal, ray_intersect, constants lines also equal SURVEY.md Appendix B.1)."""
import ctypes as C
import struct

import numpy as np


def f2u(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def kat_lines(golden, key):
    return [l.split() for l in golden["kat"] if l.split()[0] == key]


def test_seed_dist(O, golden):
    # frand.h:53-63
    for line in kat_lines(golden, "seed_dist"):
        st = C.c_uint32(int(line[1]))
        for want in line[2:]:
            v = O.lib().spo_seed_dist_next(C.byref(st))
            assert struct.unpack("<Q", struct.pack("<d", v))[0] == int(want, 16)
    # first six 15-bit outputs of seed_dist(0)
    st = C.c_uint32(0)
    got = [round(O.lib().spo_seed_dist_next(C.byref(st)) * 32767.0) for _ in range(6)]
    assert got == [38, 7719, 21238, 2437, 8855, 11797]


def test_rand_unit_vec(O, golden):
    # geom.h:164-177 through the oracle's own LCG + shared sincos
    L = O.lib()

    class V3(C.Structure):
        _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]
    L.spo_rand_unit_vec_from.restype = V3
    L.spo_rand_unit_vec_from.argtypes = [V3, C.c_double, C.c_double]
    normals = [(0, 1, 0), (0, 0, -1), (np.float32(0.6), np.float32(-0.48), np.float32(0.64))]
    states = {}
    for line in kat_lines(golden, "rand_unit_vec"):
        k, i = int(line[1]), int(line[2])
        st = states.setdefault(k, C.c_uint32(7 + k))
        r1 = L.spo_seed_dist_next(C.byref(st))
        r2 = L.spo_seed_dist_next(C.byref(st))
        v = L.spo_rand_unit_vec_from(V3(*[float(c) for c in normals[k]]), r1, r2)
        assert [f2u(v.x), f2u(v.y), f2u(v.z)] == [int(h, 16) for h in line[3:6]], (k, i)


def test_sincos_matches_reference_libm_on_every_lcg_angle(O, golden):
    """Shared sincos == the reference's std::sin/std::cos on all 2 x 32768 angles its LCG can produce."""
    L = O.lib()
    hs = hc = 1469598103934665603
    spot = {int(l[1]): l[2:] for l in kat_lines(golden, "trig")}
    M = 0xFFFFFFFFFFFFFFFF
    for k in range(32768):
        r = 1.0 * k / 32767.0
        a = np.float32(1.0 * r * np.pi * 2.0)
        b = np.float32(1.0 * r * np.pi * 0.5)
        v = [f2u(L.spo_sinf(a)), f2u(L.spo_sinf(b)), f2u(L.spo_cosf(a)), f2u(L.spo_cosf(b))]
        for u in v[:2]:
            for q in range(4):
                hs = ((hs ^ ((u >> (8 * q)) & 0xff)) * 1099511628211) & M
        for u in v[2:]:
            for q in range(4):
                hc = ((hc ^ ((u >> (8 * q)) & 0xff)) * 1099511628211) & M
        if k in spot:
            assert [f2u(a), f2u(b)] + v == [int(h, 16) for h in spot[k]]
    want = kat_lines(golden, "trig_hash")[0]
    assert f"{hs:016x}" == want[1] and f"{hc:016x}" == want[2]


def test_sincos_matches_this_libm_on_dense_sample(O):
    """glibc restatement vs the libm of the machine running the test, 200k floats across [0, 2*pi]
    (the exhaustive 1.09e9-float sweep is oracle/sincos_exhaustive.c)."""
    import math
    L = O.lib()
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    rng = np.random.default_rng(5)
    top = f2u(np.float32(2 * math.pi))
    xs = rng.integers(0, top + 1, size=200000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    for x in xs[:20000]:
        assert f2u(L.spo_sinf(x)) == f2u(libm.sinf(x)) and f2u(L.spo_cosf(x)) == f2u(libm.cosf(x)), float(x)


def test_flat_normal_and_ray_intersect(O, golden, scenes):
    tris, _ = scenes["default"]
    want = kat_lines(golden, "flat_normal")[0][1:]
    assert [f2u(x) for x in tris[0, 9:12]] == [int(h, 16) for h in want]      # python scene generator
    t0 = tris[0].copy()
    t0[9:12] = 0
    O.lib().spo_flat_normal(t0.ctypes.data_as(C.c_void_p))                       # C oracle (geom.h:192-195)
    assert [f2u(x) for x in t0[9:12]] == [int(h, 16) for h in want]
    ys = [-0.25, 0.5, -0.5, -0.1]
    for line in kat_lines(golden, "ray_intersect"):
        i = int(line[1])
        ray = np.array([0.0, ys[i], -3.0, 0.0, 0.0, 1.0], dtype=np.float32)
        pt = np.zeros(3, dtype=np.float32)
        d = O.lib().spo_ray_intersect(ray.ctypes.data_as(C.c_void_p), tris[0].ctypes.data_as(C.c_void_p), pt.ctypes.data_as(C.c_void_p))
        assert f2u(d) == int(line[2], 16)
        if len(line) > 3:
            assert [f2u(x) for x in pt] == [int(h, 16) for h in line[3:6]]


def test_vec3_rgba_and_constants(O, golden):
    L = O.lib()

    class V3(C.Structure):
        _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    class RGBA(C.Structure):
        _fields_ = [("r", C.c_uint8), ("g", C.c_uint8), ("b", C.c_uint8), ("a", C.c_uint8)]
    L.spo_vec3_rgba.restype = RGBA
    L.spo_vec3_rgba.argtypes = [V3]
    vs = [-0.1, 0.0, 0.001, 0.00196, 0.00197, 0.5, 0.998, 0.9981, 1.0, 7.0]
    want = [int(x) for x in kat_lines(golden, "vec3_RGBA")[0][1:]]
    got = []
    for v in vs:
        c = L.spo_vec3_rgba(V3(np.float32(v), np.float32(v), np.float32(v)))
        assert c.a == 0 and c.r == c.g == c.b
        got.append(c.r)
    assert got == want
    out = (C.c_float * 5)()
    inv = C.c_double()
    L.spo_consts(out, C.byref(inv))
    w = kat_lines(golden, "consts")[0][1:]
    assert [f2u(x) for x in out] == [int(h, 16) for h in w[:5]]
    assert struct.unpack("<Q", struct.pack("<d", inv.value))[0] == int(w[5], 16)
    # SURVEY.md B.1
    assert [f"{f2u(x):08x}" for x in out] == ["3e22f983", "3ea2f983", "40c90fdb", "283424dc", "5368d4a5"]


def test_philox_published_vectors(O):
    """Philox4x32-10 known answers from the Random123 distribution (kat_vectors)."""
    L = O.lib()
    A4, A2 = C.c_uint32 * 4, C.c_uint32 * 2
    for ctr, key, want in [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]:
        out = A4()
        L.spo_philox4x32_10(A4(*ctr), A2(*key), out)
        assert tuple(out) == want
