"""Camera and viewport: the producer of the hot path's input rays.

Python mirror of view::camera / view::viewport (reference src/view.h:47-132) and of the five
camera virtuals basic_renderer implements for every backend (reference
src/basic_renderer.h:32-53).  All arithmetic is float32 in the reference's operation order,
so the rays are bit-identical to camera::get_viewport (checked in tests against the compiled
reference and the C oracle).

    rays : float32 [H*W, 6] = pos.xyz dir.xyz   (24 B, geom::ray), index = i + j*W, row 0 = top
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32


def _cosf(x) -> np.float32:
    # the reference calls std::cos(float) -> glibc cosf; double cos rounded to float agrees with
    # it except in rare last-ulp cases (exact for the default angle 0)
    return F(math.cos(float(x)))


def _sinf(x) -> np.float32:
    return F(math.sin(float(x)))


class Camera:
    """view::camera (view.h:47-132): pos (0,0,-3), angle 0, focal 2 by default (view.h:76)."""

    def __init__(self, res_x: int, res_y: int, focal: float = 2.0):
        self.pos = np.array([0.0, 0.0, -3.0], dtype=F)
        self.angle = np.zeros(3, dtype=F)
        self.focal = F(focal)
        self.res_x = int(res_x)
        self.res_y = int(res_y)
        self._update_trig()

    def _update_trig(self):                       # view.h:87-92
        self.cosY, self.sinY = _cosf(self.angle[1]), _sinf(self.angle[1])
        self.cosX, self.sinX = _cosf(self.angle[0]), _sinf(self.angle[0])

    def _rX(self, x, y, z):                       # view.h:62-68
        return x, y * self.cosX + z * (-self.sinX), y * self.sinX + z * self.cosX

    def _rY(self, x, y, z):                       # view.h:54-60
        return x * self.cosY + z * self.sinY, y, x * (-self.sinY) + z * self.cosY

    def rel_move(self, x, y, z):                  # view.h:83-85
        return self._rY(*self._rX(x, y, z))

    # --- the camera virtuals of basic_renderer (basic_renderer.h:32-49)
    def set_viewport_size(self, w: int, h: int):
        self.res_x, self.res_y = int(w), int(h)

    def set_delta_mov(self, m):
        m = np.asarray(m, dtype=F)
        d = self.rel_move(m[0], m[1], m[2])
        self.pos = self.pos + np.array(d, dtype=F)

    def set_delta_rot(self, r):
        self.angle = self.angle + np.asarray(r, dtype=F)
        self._update_trig()

    def set_delta_focal(self, f):
        self.focal = F(self.focal + F(f))

    def get_viewport(self) -> np.ndarray:
        """camera::get_viewport (view.h:94-132)."""
        W, H = self.res_x, self.res_y
        x_size = F(np.float64(W) / np.float64(H))                 # :101  1.0*res_x/res_y -> real
        y_size = F(1.0)
        x_max = F(np.float64(x_size) / 2.0)
        x_step = x_size / F(W)                                    # real / size_t -> float divide
        h_x = F(np.float64(x_step) / 2.0)
        y_max = F(np.float64(y_size) / 2.0)
        y_step = y_size / F(H)
        h_y = F(np.float64(y_step) / 2.0)
        i = np.arange(W, dtype=F)
        j = np.arange(H, dtype=F)
        px = (x_max - x_step * i) - h_x                           # :111
        py = (y_max - y_step * j) - h_y
        X = np.broadcast_to(px[None, :], (H, W)).astype(F)
        Y = np.broadcast_to(py[:, None], (H, W)).astype(F)
        Z = np.zeros((H, W), dtype=F)
        dz = Z + self.focal                                       # :114 cur_pos + vec3(0,0,focal)
        dx, dy = X + F(0.0), Y + F(0.0)
        l = np.sqrt((dx * dx + dy * dy) + dz * dz)
        dx, dy, dz = dx / l, dy / l, dz / l
        X, Y, Z = self.rel_move(X, Y, Z)                          # :125-128
        dx, dy, dz = self.rel_move(dx, dy, dz)
        X, Y, Z = X + self.pos[0], Y + self.pos[1], Z + self.pos[2]  # :130-131
        rays = np.stack([X, Y, Z, dx, dy, dz], axis=-1).astype(F).reshape(H * W, 6)
        return np.ascontiguousarray(rays)
