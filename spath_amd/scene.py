"""Scene data: triangles, materials, the reference's default scene and synthetic scenes.

Layouts are the reference's tightly packed float structs (reference src/geom.h:185-190,
src/scene.h:47-50):

    triangles : float32 [N, 12]  = v0.xyz v1.xyz v2.xyz n.xyz      (48 B, geom::triangle)
    materials : float32 [N, 6]   = reflectance.rgb emittance.rgb   (24 B, scene::material)

Everything here is IEEE float32 numpy arithmetic with separately rounded operations, so the
arrays are bit-identical on every machine.
"""
from __future__ import annotations

import struct

import numpy as np

F = np.float32
SCENE_MAGIC = 0x43535053  # 'SPSC'


def flat_normals(tris: np.ndarray) -> np.ndarray:
    """n = unit((v1-v0) x (v2-v0)) in float32, same operation order as geom::flat_normal
    (reference src/geom.h:192-195, cross :143-145, unit :130-141)."""
    t = np.ascontiguousarray(tris, dtype=F).reshape(-1, 12).copy()
    a = t[:, 3:6] - t[:, 0:3]
    b = t[:, 6:9] - t[:, 0:3]
    cx = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    cy = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    cz = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    l = np.sqrt((cx * cx + cy * cy) + cz * cz)
    t[:, 9] = cx / l
    t[:, 10] = cy / l
    t[:, 11] = cz / l
    return t


def default_scene():
    """The 7-triangle scene the reference hard-codes (values from reference src/main.cpp:185-231;
    SURVEY.md Appendix C): red pyramid face, two floor triangles, a two-triangle area light,
    a two-triangle back wall."""
    p, al, wd = 20.0, 0.75, 1.0
    v = [
        [(0.0, 0.0, 1.0), (0.5, -0.5, 0.0), (-0.5, -0.5, 0.0)],
        [(p, -1.0, p), (-p, -1.0, -p), (-p, -1.0, p)],
        [(p, -1.0, p), (p, -1.0, -p), (-p, -1.0, -p)],
        [(al, 0.75, al), (-al, 0.75, al), (al, 0.75, -al)],
        [(-al, 0.75, al), (-al, 0.75, -al), (al, 0.75, -al)],
        [(1.25, 0.5, wd), (1.25, -1.0, wd), (-1.25, -1.0, wd)],
        [(1.25, 0.5, wd), (-1.25, -1.0, wd), (-1.25, 0.5, wd)],
    ]
    tris = np.zeros((7, 12), dtype=F)
    tris[:, :9] = np.asarray(v, dtype=F).reshape(7, 9)
    tris = flat_normals(tris)
    mats = np.zeros((7, 6), dtype=F)
    mats[0, 0:3] = (1.0, 0.0, 0.0)
    mats[1, 0:3] = (0.0, 1.0, 0.0)
    mats[2, 0:3] = (0.0, 0.0, 1.0)
    mats[3] = (1.0, 1.0, 1.0, 1.0, 1.0, 1.0)
    mats[4] = (1.0, 1.0, 1.0, 1.0, 1.0, 1.0)
    mats[5, 0:3] = (1.0, 1.0, 1.0)
    mats[6, 0:3] = (1.0, 1.0, 1.0)
    return tris, mats


def _hash_u32(idx: np.ndarray, stream: int, seed: int) -> np.ndarray:
    """Counter hash (murmur3 finaliser over idx, stream, seed) -> uint32."""
    x = (idx.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(stream) * np.uint64(0x85EBCA77)
         + np.uint64(seed & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    for mul in (0x85EBCA6B, 0xC2B2AE35):
        x ^= x >> np.uint64(16)
        x = (x * np.uint64(mul)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x27D4EB2F)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    return x.astype(np.uint32)


def _uniform(idx, stream, seed, lo, hi):
    """lo + u*(hi-lo) with u = 24-bit hash / 2^24, all float32 (exact conversions)."""
    u = (_hash_u32(idx, stream, seed) >> np.uint32(8)).astype(F) * F(1.0 / 16777216.0)
    return F(lo) + u * F(hi - lo)


def closed_room(n_tris: int, seed: int = 0x5CE11E, clutter_scale: float | None = None):
    """Synthetic closed scene of SURVEY.md section 8(d): a 12-triangle box around the default
    camera, a 2-triangle emissive ceiling panel and n_tris-14 small clutter triangles.  Closed,
    so every path runs all 5 closest-hit scans (nominal rays == executed scans, up to edge leaks).
    """
    if n_tris < 14:
        raise ValueError("closed_room needs at least 14 triangles")
    x0, x1, y0, y1, z0, z1 = -4.0, 4.0, -1.5, 2.5, -4.0, 4.0
    c = lambda x, y, z: (x, y, z)
    A, B, C, D = c(x0, y0, z0), c(x1, y0, z0), c(x1, y0, z1), c(x0, y0, z1)   # floor
    E, Fq, G, H = c(x0, y1, z0), c(x1, y1, z0), c(x1, y1, z1), c(x0, y1, z1)  # ceiling
    box = [
        (A, B, C), (A, C, D),       # floor
        (E, G, Fq), (E, H, G),      # ceiling
        (A, E, Fq), (A, Fq, B),     # z = z0 wall
        (D, C, G), (D, G, H),       # z = z1 wall
        (A, D, H), (A, H, E),       # x = x0 wall
        (B, Fq, G), (B, G, C),      # x = x1 wall
    ]
    ly, lh = 2.45, 1.5
    light = [
        (c(lh, ly, lh), c(-lh, ly, lh), c(lh, ly, -lh)),
        (c(-lh, ly, lh), c(-lh, ly, -lh), c(lh, ly, -lh)),
    ]
    tris = np.zeros((n_tris, 12), dtype=F)
    mats = np.zeros((n_tris, 6), dtype=F)
    tris[:12, :9] = np.asarray(box, dtype=F).reshape(12, 9)
    mats[:12, 0:3] = 0.75
    tris[12:14, :9] = np.asarray(light, dtype=F).reshape(2, 9)
    mats[12:14, :] = 1.0
    m = n_tris - 14
    if m > 0:
        if clutter_scale is None:
            clutter_scale = float(min(1.0, (10000.0 / max(n_tris, 1)) ** 0.5))
        idx = np.arange(m, dtype=np.uint64)
        ctr = np.stack([_uniform(idx, 0, seed, -1.5, 1.5), _uniform(idx, 1, seed, -1.0, 0.7),
                        _uniform(idx, 2, seed, -0.5, 2.0)], axis=1)
        r = F(0.025 * clutter_scale)
        for k in range(3):
            off = np.stack([_uniform(idx, 3 + 3 * k + a, seed, -1.0, 1.0) * r for a in range(3)], axis=1)
            tris[14:, 3 * k:3 * k + 3] = ctr + off
        for a in range(3):
            mats[14:, a] = _uniform(idx, 12 + a, seed, 0.2, 0.9)
    tris = flat_normals(tris)
    return tris, mats


def open_clutter(n_tris: int, seed: int = 7):
    """Small open test scene: the default scene's floor/light plus random triangles; paths may
    escape, exercising the miss path."""
    if n_tris < 7:
        raise ValueError("open_clutter needs at least the 7 default triangles")
    base_t, base_m = default_scene()
    m = n_tris - 7
    tris = np.zeros((7 + m, 12), dtype=F)
    mats = np.zeros((7 + m, 6), dtype=F)
    tris[:7], mats[:7] = base_t, base_m
    if m:
        idx = np.arange(m, dtype=np.uint64)
        ctr = np.stack([_uniform(idx, 0, seed, -1.2, 1.2), _uniform(idx, 1, seed, -0.9, 0.6),
                        _uniform(idx, 2, seed, -1.0, 0.9)], axis=1)
        for k in range(3):
            off = np.stack([_uniform(idx, 3 + 3 * k + a, seed, -0.2, 0.2) for a in range(3)], axis=1)
            tris[7:, 3 * k:3 * k + 3] = ctr + off
        for a in range(3):
            mats[7:, a] = _uniform(idx, 12 + a, seed, 0.1, 1.0)
        tris = flat_normals(tris)
    return tris, mats


def write_scene(path, tris: np.ndarray, mats: np.ndarray) -> None:
    """Scene file read by oracle/ref_driver.cpp and the headless CLI: 'SPSC', n, tris, mats."""
    tris = np.ascontiguousarray(tris, dtype=F).reshape(-1, 12)
    mats = np.ascontiguousarray(mats, dtype=F).reshape(-1, 6)
    assert tris.shape[0] == mats.shape[0]
    with open(path, "wb") as f:
        f.write(struct.pack("<II", SCENE_MAGIC, tris.shape[0]))
        f.write(tris.tobytes())
        f.write(mats.tobytes())


def read_scene(path):
    with open(path, "rb") as f:
        magic, n = struct.unpack("<II", f.read(8))
        if magic != SCENE_MAGIC:
            raise ValueError(f"{path}: not a spath scene file")
        tris = np.frombuffer(f.read(n * 48), dtype=F).reshape(n, 12).copy()
        mats = np.frombuffer(f.read(n * 24), dtype=F).reshape(n, 6).copy()
    return tris, mats
