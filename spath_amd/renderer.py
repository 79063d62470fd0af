"""Python mirror of the reference's renderer plugin interface for the HIP backend.

`Renderer` has the eight virtuals of scene::renderer (reference src/renderer.h:24-36);
`BasicRenderer` supplies the five camera ones exactly as basic_renderer does for every backend
(reference src/basic_renderer.h:25-54); `HipRenderer` implements get_description / render_flat /
render on top of the C ABI (include/spath_hip.h), i.e. it is the Python twin of the C++ adapter
spath_amd/host/hip_renderer.cpp.  `get(w, h)` mirrors the per-backend factory
`X_renderer::get(w, h)` (reference src/cpu_renderer.h:23-25).

Argument meaning and error behaviour follow the reference: render calls are synchronous, borrow
the caller's arrays only for the duration of the call, size the output bitmap themselves
(cpu_renderer.cpp:120-122) and raise (the reference throws std::runtime_error) on device failure.
"""
from __future__ import annotations

import numpy as np

from . import capi
from .view import Camera


class Bitmap:
    """scene::bitmap (reference src/scene.h:41-45): res_x, res_y, values[res_x*res_y] RGBA8."""

    def __init__(self):
        self.res_x = 0
        self.res_y = 0
        self.values = np.zeros((0, 4), dtype=np.uint8)

    def image(self) -> np.ndarray:
        return self.values.reshape(self.res_y, self.res_x, 4)


class Viewport:
    """view::viewport (reference src/view.h:28-31): res_x, res_y, rays[res_x*res_y]."""

    def __init__(self, res_x=0, res_y=0, rays=None):
        self.res_x, self.res_y = res_x, res_y
        self.rays = rays if rays is not None else np.zeros((0, 6), dtype=np.float32)


class Renderer:
    """scene::renderer (reference src/renderer.h:24-36)."""

    def get_description(self) -> str: raise NotImplementedError
    def set_viewport_size(self, w: int, h: int): raise NotImplementedError
    def set_delta_mov(self, m): raise NotImplementedError
    def set_delta_rot(self, r): raise NotImplementedError
    def set_delta_focal(self, f: float): raise NotImplementedError
    def get_viewport(self, vp: Viewport): raise NotImplementedError
    def render_flat(self, vp, tris, mats, n_tris, n_samples, out: Bitmap): raise NotImplementedError
    def render(self, vp, tris, mats, n_tris, n_samples, out: Bitmap): raise NotImplementedError


class BasicRenderer(Renderer):
    """basic_renderer (reference src/basic_renderer.h:25-54): owns the camera."""

    def __init__(self, x: int, y: int):
        self.vc = Camera(x, y)

    def set_viewport_size(self, w, h): self.vc.set_viewport_size(w, h)
    def set_delta_mov(self, m): self.vc.set_delta_mov(m)
    def set_delta_rot(self, r): self.vc.set_delta_rot(r)
    def set_delta_focal(self, f): self.vc.set_delta_focal(f)

    def get_viewport(self, vp: Viewport):
        vp.res_x, vp.res_y = self.vc.res_x, self.vc.res_y
        vp.rays = self.vc.get_viewport()


class HipRenderer(BasicRenderer):
    """The MI355X backend behind the reference's plugin interface."""

    def __init__(self, x: int, y: int, device: int = 0, seed: int = 1, flags: int = 0):
        super().__init__(x, y)
        self.ctx = capi.Context(device)      # raises like cl_r/vk_r constructors do on init failure
        self.seed = seed
        self.flags = flags
        self._scene_key = None
        self.last_stats = None

    def get_description(self) -> str:
        return self.ctx.description

    def _upload_scene(self, tris, mats, n_tris):
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 12)[:n_tris]
        mats = np.ascontiguousarray(mats, dtype=np.float32).reshape(-1, 6)[:n_tris]
        # the reference's GPU peer re-uploads every frame (cl_renderer.cpp:210-214); upload only on change
        key = (n_tris, hash(tris.tobytes()), hash(mats.tobytes()))
        if key != self._scene_key:
            self.ctx.set_scene(tris, mats)
            self._scene_key = key

    def _render(self, vp, tris, mats, n_tris, n_samples, out, mode):
        self._upload_scene(tris, mats, n_tris)
        out.res_x, out.res_y = vp.res_x, vp.res_y          # cpu_renderer.cpp:120-122
        out.values = self.ctx.render(vp.rays, vp.res_x, vp.res_y, n_samples, seed=self.seed, mode=mode, flags=self.flags)
        self.last_stats = self.ctx.stats()

    def render_flat(self, vp, tris, mats, n_tris, n_samples, out):
        self._render(vp, tris, mats, n_tris, max(int(n_samples), 1), out, capi.MODE_FLAT)

    def render(self, vp, tris, mats, n_tris, n_samples, out):
        self._render(vp, tris, mats, n_tris, n_samples, out, capi.MODE_PT)

    def render_own_viewport(self, tris, mats, n_tris, n_samples, out: Bitmap, flat: bool = False):
        """get_viewport + render(_flat) without moving rays over PCIe: the viewport is generated on the device
        from this renderer's own camera (bit-identical to get_viewport)."""
        self._upload_scene(tris, mats, n_tris)
        out.res_x, out.res_y = self.vc.res_x, self.vc.res_y
        out.values = self.ctx.render_camera(self.vc, max(int(n_samples), 1), seed=self.seed,
                                            mode=capi.MODE_FLAT if flat else capi.MODE_PT, flags=self.flags)
        self.last_stats = self.ctx.stats()

    def close(self):
        self.ctx.close()


def get(w: int, h: int, **kw) -> Renderer:
    """hip_renderer::get(w, h) -- peer of cpu_renderer::get (reference src/cpu_renderer.cpp:205-209)."""
    return HipRenderer(w, h, **kw)
