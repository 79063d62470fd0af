"""TEST INFRASTRUCTURE ONLY -- ctypes access to the parity oracle.

Loads oracle/liboracle.so (this repo's plain-C restatement of the reference CPU renderer,
oracle/spath_oracle.c) and, where present, runs oracle/_ref/spath_ref (the unmodified
reference CPU backend compiled behind oracle/ref_driver.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under spath_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_BIN = os.path.join(HERE, "_ref", "spath_ref")

F = np.float32
_lib = None


def build(force: bool = False) -> None:
    """Compile the C restatement (and the reference binary when /root/reference is there)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "spath_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        fp, u8p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_void_p
        L.spo_render_flat.argtypes = [vp, C.c_size_t, C.c_size_t, vp, vp, C.c_size_t, vp]
        L.spo_render_mt.argtypes = [vp, C.c_size_t, C.c_size_t, vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, vp]
        L.spo_render_counter.argtypes = [vp, C.c_size_t, C.c_size_t, vp, vp, C.c_size_t, C.c_size_t, C.c_uint64,
                                         C.c_int, vp, vp, C.POINTER(C.c_uint64)]
        L.spo_seed_dist_next.restype = C.c_double
        L.spo_seed_dist_next.argtypes = [C.POINTER(C.c_uint32)]
        L.spo_sinf.restype = C.c_float
        L.spo_sinf.argtypes = [C.c_float]
        L.spo_cosf.restype = C.c_float
        L.spo_cosf.argtypes = [C.c_float]
        L.spo_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.spo_counter_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.spo_device_math_batch.argtypes = [C.c_int, vp, C.c_size_t, vp, C.c_int]
        L.spo_ray_intersect.restype = C.c_float
        L.spo_ray_intersect.argtypes = [vp, vp, vp]
        L.spo_closest_hit.restype = C.c_int
        L.spo_closest_hit.argtypes = [vp, vp, C.c_size_t, C.c_int, fp, vp]
        L.spo_closest_hit_batch.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp]
        L.spo_flat_normal.argtypes = [vp]
        L.spo_consts.argtypes = [fp, C.POINTER(C.c_double)]
        L.spo_camera_init.argtypes = [vp, C.c_size_t, C.c_size_t]
        L.spo_camera_get_viewport.argtypes = [vp, vp]
        _lib = L
    return _lib


class _V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _prep(rays, tris, mats):
    rays = np.ascontiguousarray(rays, dtype=F).reshape(-1, 6)
    tris = np.ascontiguousarray(tris, dtype=F).reshape(-1, 12)
    mats = np.ascontiguousarray(mats, dtype=F).reshape(-1, 6)
    assert tris.shape[0] == mats.shape[0]
    return rays, tris, mats


def render_flat(rays, w, h, tris, mats):
    rays, tris, mats = _prep(rays, tris, mats)
    out = np.zeros((w * h, 4), dtype=np.uint8)
    lib().spo_render_flat(_p(rays), w, h, _p(tris), _p(mats), tris.shape[0], _p(out))
    return out


def render_mt(rays, w, h, tris, mats, n_samples, threads, workers=None):
    """cpu_renderer's render() with `threads` simulated reference threads."""
    rays, tris, mats = _prep(rays, tris, mats)
    out = np.zeros((w * h, 4), dtype=np.uint8)
    workers = workers or min(threads, os.cpu_count() or 1)
    lib().spo_render_mt(_p(rays), w, h, _p(tris), _p(mats), tris.shape[0], n_samples, threads, workers, _p(out))
    return out


def render_counter(rays, tris, mats, n_samples, seed, pix0=0, npix=None, workers=None):
    """Same integrator with the counter RNG.  Returns (rgba [npix,4] u8, accum [npix,3] f32, scans)."""
    rays, tris, mats = _prep(rays, tris, mats)
    if npix is None:
        npix = rays.shape[0] - pix0
    out = np.zeros((npix, 4), dtype=np.uint8)
    acc = np.zeros((npix, 3), dtype=F)
    scans = C.c_uint64(0)
    workers = workers or (os.cpu_count() or 1)
    lib().spo_render_counter(_p(rays), pix0, npix, _p(tris), _p(mats), tris.shape[0], n_samples,
                             C.c_uint64(seed), workers, _p(out), _p(acc), C.byref(scans))
    return out, acc, int(scans.value)


MATH_OUT = {0: (np.float32, 2), 2: (np.float64, 2), 3: (np.float32, 3), 4: (np.float32, 1), 5: (np.uint32, 1)}


def device_math(what, inp, n):
    """Batch counterparts of the device functions sphip_selftest_device evaluates (same `what` codes, same layouts)."""
    dt, k = MATH_OUT[what]
    inp = np.ascontiguousarray(inp)
    out = np.zeros(n * k, dtype=dt)
    lib().spo_device_math_batch(what, _p(inp), n, _p(out), os.cpu_count() or 1)
    return out


def closest_hits(rays, tris, src_idx=None):
    """cpu_renderer.cpp:36-49 for a batch of rays: (idx [n] i32, d [n] f32)."""
    rays = np.ascontiguousarray(rays, dtype=F).reshape(-1, 6)
    tris = np.ascontiguousarray(tris, dtype=F).reshape(-1, 12)
    idx = np.zeros(rays.shape[0], dtype=np.int32)
    d = np.zeros(rays.shape[0], dtype=F)
    src = None if src_idx is None else np.ascontiguousarray(src_idx, dtype=np.int32)
    lib().spo_closest_hit_batch(_p(rays), rays.shape[0], _p(tris), tris.shape[0], _p(src) if src is not None else None, _p(idx), _p(d))
    return idx, d


def viewport(w, h, moves=()):
    """camera::get_viewport through the C restatement.  moves: sequence of ('mov'|'rot', (x,y,z)) / ('focal', f)."""
    L = lib()
    cam = (C.c_uint8 * 128)()
    L.spo_camera_init(cam, w, h)
    for kind, val in moves:
        if kind == "mov":
            L.spo_camera_delta_mov.argtypes = [C.c_void_p, _V3]
            L.spo_camera_delta_mov(cam, _V3(*val))
        elif kind == "rot":
            L.spo_camera_delta_rot.argtypes = [C.c_void_p, _V3]
            L.spo_camera_delta_rot(cam, _V3(*val))
        elif kind == "focal":
            L.spo_camera_delta_focal.argtypes = [C.c_void_p, C.c_float]
            L.spo_camera_delta_focal(cam, C.c_float(val))
        else:
            raise ValueError(kind)
    rays = np.zeros((w * h, 6), dtype=F)
    L.spo_camera_get_viewport(cam, _p(rays))
    return rays


# ----------------------------------------------------------------------------- the real reference
def have_ref() -> bool:
    return os.path.isfile(REF_BIN) and os.access(REF_BIN, os.X_OK)


def _cam_args(moves):
    args = []
    for kind, val in moves:
        if kind in ("mov", "rot"):
            args += [kind] + [repr(float(v)) for v in val]
        else:
            args += [kind, repr(float(val))]
    return args


def ref_run(mode, w, h, spp=1, tris=None, mats=None, threads=8, moves=(), rays=None, return_info=False):
    """Run the compiled reference.  mode: render | flat | viewport."""
    from spath_amd.scene import write_scene
    with tempfile.TemporaryDirectory() as td:
        scene = "-"
        if tris is not None:
            scene = os.path.join(td, "scene.bin")
            write_scene(scene, tris, mats)
        outp = os.path.join(td, "out.bin")
        cmd = [REF_BIN, mode, scene, str(w), str(h), str(spp), outp] + _cam_args(moves)
        if rays is not None:
            rp = os.path.join(td, "rays.bin")
            np.ascontiguousarray(rays, dtype=F).tofile(rp)
            cmd += ["rays", rp]
        # a profiler wrapped around the caller (rocprofv3 LD_PRELOADs its tool) must not follow into the CPU child
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF"))}
        env["ORACLE_THREADS"] = str(threads)
        pr = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True)
        if mode == "viewport":
            res = np.fromfile(outp, dtype=F).reshape(w * h, 6)
        else:
            res = np.fromfile(outp, dtype=np.uint8).reshape(w * h, 4)
        if return_info:
            info = {}
            for line in pr.stderr.decode(errors="replace").splitlines():
                if line.startswith('{"mode"'):
                    info = json.loads(line)
            return res, info
        return res


def ref_kat():
    with tempfile.TemporaryDirectory() as td:
        outp = os.path.join(td, "kat.txt")
        subprocess.run([REF_BIN, "kat", "-", "0", "0", "0", outp], check=True)
        return open(outp).read()


def fnv1a64(data: bytes) -> str:
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"
