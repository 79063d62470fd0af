"""ctypes binding of the C ABI in include/spath_hip.h (libspath_hip.so).

This is the only way Python reaches the HIP kernels.  There is no CPU fallback: if the shared
library is missing or the GPU cannot be initialised the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libspath_hip.so")

MODE_FLAT = 0
MODE_PT = 1
KERNEL_AUTO = 0
FLAG_PRIMARY_REUSE = 0x100
FLAG_ACCEL = 0x200          # opt-in linear BVH; not the brute-force path


def flag_chunks(n: int) -> int:
    """flags bits 16..23: number of sample chunks of a path-traced launch (0 = library's choice, 1 = never split)."""
    if not 0 <= n <= 255:
        raise ValueError("chunks must be in [0, 255]")
    return n << 16

# every entry point include/spath_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = (
    "sphip_create", "sphip_destroy", "sphip_last_error", "sphip_description", "sphip_abi_version",
    "sphip_kernel_name", "sphip_set_scene", "sphip_render", "sphip_set_scene_device",
    "sphip_render_device", "sphip_closest_hit_device", "sphip_get_stats", "sphip_viewport_device", "sphip_render_camera",
    "sphip_create_multi", "sphip_device_count", "sphip_plan_tile_rows", "sphip_plan_shard", "sphip_selftest_device",
    "sphip_kernel_available", "sphip_selftest_stage1", "sphip_build_info",
)
GATHER_NONE, GATHER_RCCL, GATHER_PEER = 0, 1, 2


class Shard(C.Structure):
    _fields_ = [("pixel_base", C.c_uint64), ("tile_px", C.c_uint64), ("tile_stride_px", C.c_uint64)]


class CameraArgs(C.Structure):
    """sphip_camera: view::camera state with the trig values the reference computes on the host (view.h:77-80)."""
    _fields_ = [("pos", C.c_float * 3), ("cos_y", C.c_float), ("sin_y", C.c_float), ("cos_x", C.c_float), ("sin_x", C.c_float),
                ("focal", C.c_float), ("res_x", C.c_uint32), ("res_y", C.c_uint32)]

    @classmethod
    def from_camera(cls, cam):
        """cam: spath_amd.view.Camera"""
        return cls((C.c_float * 3)(*[float(x) for x in cam.pos]), float(cam.cosY), float(cam.sinY), float(cam.cosX), float(cam.sinX),
                   float(cam.focal), int(cam.res_x), int(cam.res_y))


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("upload_ms", C.c_double), ("download_ms", C.c_double),
                ("scans_executed", C.c_uint64), ("n_tris", C.c_uint64), ("n_pixels", C.c_uint64),
                ("kernel_variant", C.c_uint32), ("n_launches", C.c_uint32),
                ("n_devices", C.c_uint32), ("gather_kind", C.c_uint32), ("gather_ms", C.c_double), ("kernel_ms_min", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SpathHipError(RuntimeError):
    """A non-zero status from the C ABI (the C++ adapter throws std::runtime_error at the same point)."""


_lib = None


def load():
    """Load libspath_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpathHipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                            "there is no CPU fallback for the HIP path")
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.sphip_abi_version.restype = C.c_int
    L.sphip_kernel_name.restype = C.c_char_p
    L.sphip_kernel_name.argtypes = [C.c_int]
    L.sphip_create.restype = C.c_int
    L.sphip_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.sphip_destroy.restype = None
    L.sphip_destroy.argtypes = [vp]
    L.sphip_last_error.restype = C.c_char_p
    L.sphip_last_error.argtypes = [vp]
    L.sphip_description.restype = C.c_char_p
    L.sphip_description.argtypes = [vp]
    L.sphip_set_scene.restype = C.c_int
    L.sphip_set_scene.argtypes = [vp, vp, vp, sz]
    L.sphip_render.restype = C.c_int
    L.sphip_render.argtypes = [vp, vp, sz, sz, sz, C.c_uint64, C.c_int, C.c_int, vp, vp]
    L.sphip_set_scene_device.restype = C.c_int
    L.sphip_set_scene_device.argtypes = [vp, vp, vp, sz, vp]
    L.sphip_render_device.restype = C.c_int
    L.sphip_render_device.argtypes = [vp, vp, sz, C.POINTER(Shard), sz, sz, C.c_uint64, C.c_int, C.c_int, vp, vp, vp]
    L.sphip_closest_hit_device.restype = C.c_int
    L.sphip_closest_hit_device.argtypes = [vp, vp, sz, vp, C.c_int, vp, vp, vp]
    L.sphip_viewport_device.restype = C.c_int
    L.sphip_viewport_device.argtypes = [vp, C.POINTER(CameraArgs), vp, vp]
    L.sphip_render_camera.restype = C.c_int
    L.sphip_render_camera.argtypes = [vp, C.POINTER(CameraArgs), sz, C.c_uint64, C.c_int, C.c_int, vp, vp]
    L.sphip_get_stats.restype = C.c_int
    L.sphip_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.sphip_selftest_device.restype = C.c_int
    L.sphip_selftest_device.argtypes = [vp, C.c_int, vp, sz, vp]
    # older builds (A/B libraries of earlier rounds) lack the next two: bind them when present
    if hasattr(L, "sphip_kernel_available"):
        L.sphip_kernel_available.restype = C.c_int
        L.sphip_kernel_available.argtypes = [C.c_int]
    if hasattr(L, "sphip_build_info"):
        L.sphip_build_info.restype = C.c_char_p
        L.sphip_build_info.argtypes = []
    if hasattr(L, "sphip_selftest_stage1"):
        L.sphip_selftest_stage1.restype = C.c_int
        L.sphip_selftest_stage1.argtypes = [vp, vp, sz, vp, vp, vp, C.POINTER(C.c_uint32)]
    L.sphip_create_multi.restype = C.c_int
    L.sphip_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.sphip_device_count.restype = C.c_int
    L.sphip_device_count.argtypes = [vp]
    L.sphip_plan_tile_rows.restype = C.c_int
    L.sphip_plan_tile_rows.argtypes = [sz, C.c_int]
    L.sphip_plan_shard.restype = C.c_int
    L.sphip_plan_shard.argtypes = [sz, sz, C.c_int, sz, C.c_int, C.POINTER(Shard), C.POINTER(sz)]
    _lib = L
    return L


def kernel_variants():
    """{name: id} of the scan-kernel variants the library exposes (id 0 = auto)."""
    L = load()
    out, i = {}, 0
    while True:
        n = L.sphip_kernel_name(i)
        if n is None:
            break
        out[n.decode()] = i
        i += 1
    return out


def build_source_hash() -> str:
    """The source/flags hash compiled into the loaded library ("unknown" for hand-made builds and libraries of earlier rounds)."""
    L = load()
    if not hasattr(L, "sphip_build_info"):
        return "unknown"
    return L.sphip_build_info().decode().split("src=")[-1]


def available_variants():
    """ids (> 0) of the kernel variants THIS build of the library carries (the shipped build leaves the earlier filter
    generations out; a -DSP_ALL_VARIANTS build has them all)."""
    L = load()
    ids = sorted(v for v in kernel_variants().values() if v > 0)
    if not hasattr(L, "sphip_kernel_available"):
        return ids
    return [v for v in ids if L.sphip_kernel_available(v)]


def plan_tile_rows(height: int, n_devices: int) -> int:
    """The library's row-tile height for a frame of `height` rows on n_devices GPUs (pure host arithmetic)."""
    return load().sphip_plan_tile_rows(height, n_devices)


def plan_shard(width: int, height: int, n_devices: int, tile_rows: int, rank: int):
    """((pixel_base, tile_px, tile_stride_px), n_rays) of device `rank` in the library's round-robin row-tile plan."""
    sh, n = Shard(), C.c_size_t(0)
    rc = load().sphip_plan_shard(width, height, n_devices, tile_rows, rank, C.byref(sh), C.byref(n))
    if rc != 0:
        raise SpathHipError(f"sphip_plan_shard: bad arguments [{rc}]")
    return (sh.pixel_base, sh.tile_px, sh.tile_stride_px), n.value


class Context:
    """One sphip_t: a device (or, from Context.multi, several), its cached buffers and its scene."""

    def __init__(self, device: int = 0, _handle=None):
        self._L = load()
        if _handle is not None:
            self._h, self.device = _handle, None
            return
        h = C.c_void_p()
        rc = self._L.sphip_create(device, C.byref(h))
        if rc != 0:
            raise SpathHipError(f"sphip_create({device}) failed [{rc}]: {self._L.sphip_last_error(None).decode()}")
        self._h = h
        self.device = device

    @classmethod
    def multi(cls, device_ids=None):
        """sphip_create_multi: every listed device behind one context (None: all visible devices / SPATH_HIP_DEVICES).
        Only the host-pointer calls (set_scene, render, render_camera, stats) work on it."""
        L = load()
        h = C.c_void_p()
        if device_ids is None:
            rc = L.sphip_create_multi(None, 0, C.byref(h))
        else:
            arr = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
            rc = L.sphip_create_multi(arr, len(device_ids), C.byref(h))
        if rc != 0:
            raise SpathHipError(f"sphip_create_multi({device_ids}) failed [{rc}]: {L.sphip_last_error(None).decode()}")
        return cls(_handle=h)

    @property
    def device_count(self) -> int:
        return self._L.sphip_device_count(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.sphip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise SpathHipError(f"{what} failed [{rc}]: {self._L.sphip_last_error(self._h).decode()}")

    @property
    def description(self) -> str:
        return self._L.sphip_description(self._h).decode()

    # host-pointer path -------------------------------------------------------------------------
    def set_scene(self, tris, mats):
        import numpy as np
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 12)
        mats = np.ascontiguousarray(mats, dtype=np.float32).reshape(-1, 6)
        if tris.shape[0] != mats.shape[0]:
            raise ValueError("one material per triangle")
        self._check(self._L.sphip_set_scene(self._h, tris.ctypes.data, mats.ctypes.data, tris.shape[0]), "sphip_set_scene")

    def render(self, rays, w, h, n_samples, seed=1, mode=MODE_PT, flags=0, want_accum=False):
        import numpy as np
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        if rays.shape[0] != w * h:
            raise ValueError("rays must hold w*h entries")
        out = np.zeros((w * h, 4), dtype=np.uint8)
        acc = np.zeros((w * h, 3), dtype=np.float32) if want_accum else None
        self._check(self._L.sphip_render(self._h, rays.ctypes.data, w, h, n_samples, seed, mode, flags,
                                         out.ctypes.data, acc.ctypes.data if want_accum else None), "sphip_render")
        return (out, acc) if want_accum else out

    # device-resident path ----------------------------------------------------------------------
    def set_scene_device(self, d_tris: int, d_mats: int, n_tris: int, stream: int = 0):
        self._check(self._L.sphip_set_scene_device(self._h, d_tris, d_mats, n_tris, stream), "sphip_set_scene_device")

    def render_device(self, d_rays: int, n_rays: int, n_samples: int, d_out_rgba: int, *, seed=1, mode=MODE_PT,
                      flags=0, shard=None, image_width=0, d_out_accum: int = 0, stream: int = 0):
        sh = None
        if shard is not None:
            sh = C.byref(Shard(*[int(v) for v in shard]))
        self._check(self._L.sphip_render_device(self._h, d_rays, n_rays, sh, image_width, n_samples, seed, mode, flags,
                                                d_out_rgba, d_out_accum or None, stream or None), "sphip_render_device")

    def closest_hit_device(self, d_rays: int, n_rays: int, d_out_idx: int, d_out_dist: int, *, d_src_idx: int = 0,
                           flags=0, stream: int = 0):
        self._check(self._L.sphip_closest_hit_device(self._h, d_rays, n_rays, d_src_idx or None, flags,
                                                     d_out_idx, d_out_dist, stream or None), "sphip_closest_hit_device")

    def viewport_device(self, cam, d_rays_out: int, stream: int = 0):
        """camera::get_viewport on the device (cam: spath_amd.view.Camera); writes res_x*res_y*6 floats."""
        ca = CameraArgs.from_camera(cam)
        self._check(self._L.sphip_viewport_device(self._h, C.byref(ca), d_rays_out, stream or None), "sphip_viewport_device")

    def render_camera(self, cam, n_samples, seed=1, mode=MODE_PT, flags=0, want_accum=False):
        """get_viewport + render in one call; the rays never leave the device."""
        import numpy as np
        ca = CameraArgs.from_camera(cam)
        n = cam.res_x * cam.res_y
        out = np.zeros((n, 4), dtype=np.uint8)
        acc = np.zeros((n, 3), dtype=np.float32) if want_accum else None
        self._check(self._L.sphip_render_camera(self._h, C.byref(ca), n_samples, seed, mode, flags, out.ctypes.data,
                                                acc.ctypes.data if want_accum else None), "sphip_render_camera")
        return (out, acc) if want_accum else out

    SELFTEST_OUT = {0: ("float32", 2), 1: ("float32", 1), 2: ("float64", 2), 3: ("float32", 3), 4: ("float32", 1), 5: ("uint32", 1), 6: ("float32", 2)}

    def selftest(self, what: int, inp, n: int):
        """sphip_selftest_device (test-only): one device function of the path on n caller-supplied inputs."""
        import numpy as np
        dt, k = self.SELFTEST_OUT[what]
        inp = np.ascontiguousarray(inp)
        out = np.zeros(n * k, dtype=dt)
        self._check(self._L.sphip_selftest_device(self._h, what, inp.ctypes.data, n, out.ctypes.data), "sphip_selftest_device")
        return out

    def selftest_stage1(self, rays, per_triangle=True):
        """sphip_selftest_stage1 (test-only): stage 1 of the default scan alone for the given rays (padded to a multiple of 64)
        against the context's scene.  Returns (group_survives, tri_survives, order): both [ray, stream position] bool -- the
        position's OCTET (its group of four and the partner group two further on) survives the bound of the octet / the triangle itself survives
        the per-pair form of the test
        (None unless per_triangle) -- and order[stream position] = triangle index (n_tris = padding)."""
        import numpy as np
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        n_pad = (n + 63) // 64 * 64
        if n_pad != n:
            rays = np.concatenate([rays, np.repeat(rays[-1:], n_pad - n, axis=0)])
        tiles = C.c_uint32(0)
        self._check(self._L.sphip_selftest_stage1(self._h, None, 0, None, None, None, C.byref(tiles)), "sphip_selftest_stage1")
        t, T, octets = tiles.value & 0xFFFFF, (tiles.value >> 20) & 0x7FF, bool(tiles.value >> 31)
        G = T // 4                                                       # groups of four per tile
        W = max(1, T // 512) if octets else T // 256                     # words per ray block: 2 (octets) or 4 (groups) bits per fragment
        words = np.zeros(n_pad * t * 2 * W, dtype=np.uint32)
        tri = np.zeros(n_pad * t * 2 * (T // 64), dtype=np.uint32) if per_triangle else None
        order = np.zeros(t * T, dtype=np.int32)
        self._check(self._L.sphip_selftest_stage1(self._h, rays.ctypes.data, n_pad, words.ctypes.data, tri.ctypes.data if per_triangle else None,
                                                  order.ctypes.data, C.byref(tiles)), "sphip_selftest_stage1")
        nb = n_pad // 64
        # words[(64 b + l), tile, rb, w], bit 31 - k, ray 64 b + (l & 31) + 32 rb:
        #   group bits:  k = 4 f + j <-> group 8 (8 w + f) + 2 j + (l >> 5) = 64 w + 2 k + (l >> 5)
        #   octet bits:  k = 2 f + q <-> groups g = 8 (16 w + f) + 4 q + (l >> 5) = 128 w + 4 k + (l >> 5) and g + 2
        w = words.reshape(nb, 2, 32, t, 2, W)                            # [block, half, column, tile, rb, word]
        bits = ((w[..., None] >> (31 - np.arange(32, dtype=np.uint32))) & 1).astype(bool)     # [..., k]
        wi, k = np.meshgrid(np.arange(W), np.arange(32), indexing="ij")
        surv = np.zeros((nb, 64, t, G), dtype=bool)                      # [block, ray in block, tile, group of four]
        for hh in range(2):
            g0 = ((128 * wi + 4 * k if octets else 64 * wi + 2 * k) + hh).reshape(-1)
            keep = g0 + (2 if octets else 0) < G                         # (a 256-triangle tile with octet bits uses the upper half of its one word)
            for rb in range(2):
                b = bits[:, hh, :, :, rb].reshape(nb, 32, t, W * 32)[..., keep]
                surv[:, 32 * rb:32 * rb + 32, :, g0[keep]] = b
                if octets:
                    surv[:, 32 * rb:32 * rb + 32, :, g0[keep] + 2] = b
        surv = np.repeat(surv.reshape(n_pad, t * G), 4, axis=1)          # per stream position
        tsurv = None
        if per_triangle:
            # tri[(64 b + l), tile, rb, q]: bit 31 - (16 (f & 1) + 4 j + i), f = 2 q + (bit >= 16) <-> triangle 32 f + 8 j + 4 (l >> 5) + i
            Q = T // 64
            tw = tri.reshape(nb, 2, 32, t, 2, Q)
            tb = ((tw[..., None] >> (31 - np.arange(32, dtype=np.uint32))) & 1).astype(bool)   # [block, half, column, tile, rb, q, bit]
            tsurv = np.zeros((nb, 64, t, T), dtype=bool)
            q, bit = np.meshgrid(np.arange(Q), np.arange(32), indexing="ij")
            f, j, i = 2 * q + bit // 16, (bit % 16) // 4, bit % 4
            for hh in range(2):
                pos = (32 * f + 8 * j + 4 * hh + i).reshape(-1)
                for rb in range(2):
                    tsurv[:, 32 * rb:32 * rb + 32, :, pos] = tb[:, hh, :, :, rb].reshape(nb, 32, t, Q * 32)
            tsurv = tsurv.reshape(n_pad, t * T)[:n]
        return surv[:n], tsurv, order

    def stats(self) -> dict:
        s = Stats()
        self._check(self._L.sphip_get_stats(self._h, C.byref(s)), "sphip_get_stats")
        return s.as_dict()
