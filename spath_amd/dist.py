"""Pixel-row-tile sharding of the framebuffer across the GPUs of one node, and its reassembly.

The reference has no multi-device code; pixels are independent
(reference src/cpu_renderer.cpp:70-79) and the HIP path keys its RNG by global pixel index, so the
image does not depend on the partition.  One process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm) renders its tiles; the only exchange step is one gather of RGBA8 tiles to rank 0.

Tiles of `tile_rows` image rows are dealt round-robin (tile t -> rank t mod G) so that cheap (sky)
and expensive rows spread evenly in open scenes.
"""
from __future__ import annotations

import numpy as np


def balanced_tile_rows(height: int, world_size: int, max_rows: int = 8) -> int:
    """Largest tile height <= max_rows that deals every rank the same number of full tiles (so that equal-cost pixels give
    equal shards); max_rows if there is none.  1080 rows on 8 ranks: 5 (216 tiles, 27 each) instead of 8 (17/16 tiles)."""
    for tr in range(max_rows, 0, -1):
        if height % tr == 0 and (height // tr) % world_size == 0:
            return tr
    return max_rows


class RowTilePlan:
    def __init__(self, width: int, height: int, world_size: int, tile_rows: int = 8):
        if min(width, height, world_size, tile_rows) < 1:
            raise ValueError("bad plan")
        self.w, self.h, self.g, self.tile_rows = width, height, world_size, tile_rows
        self.tile_px = tile_rows * width
        self.n_tiles = -(-height // tile_rows)
        self.npix = width * height
        self._ids = {}            # (rank, device) -> global pixel ids as a torch tensor, built once

    def tiles_of(self, rank: int):
        return list(range(rank, self.n_tiles, self.g))

    def n_rays(self, rank: int) -> int:
        n = 0
        for t in self.tiles_of(rank):
            n += min(self.tile_px, self.npix - t * self.tile_px)
        return n

    def max_rays(self) -> int:
        return max(self.n_rays(r) for r in range(self.g))

    def shard(self, rank: int):
        """(pixel_base, tile_px, tile_stride_px) of include/spath_hip.h's sphip_shard."""
        return (rank * self.tile_px, self.tile_px, self.g * self.tile_px)

    def pixel_ids(self, rank: int) -> np.ndarray:
        """Global pixel index of each local ray, in local order (== the sphip_shard formula)."""
        ids = [np.arange(t * self.tile_px, min((t + 1) * self.tile_px, self.npix), dtype=np.int64)
               for t in self.tiles_of(rank)]
        return np.concatenate(ids) if ids else np.zeros(0, dtype=np.int64)

    def pixel_ids_on(self, rank: int, device):
        """pixel_ids(rank) as an int64 tensor on `device`, cached: the un-permute of every frame reuses it."""
        import torch
        key = (rank, str(device))
        if key not in self._ids:
            self._ids[key] = torch.from_numpy(self.pixel_ids(rank)).to(device)
        return self._ids[key]

    def assemble(self, gathered, out=None):
        """gathered: [G, max_rays, C] (numpy or torch), or a list of G [max_rays, C] tensors -> [H*W, C] image in global pixel order."""
        import torch
        is_np = isinstance(gathered, np.ndarray)
        g = torch.from_numpy(gathered) if is_np else gathered
        first = g[0]
        if out is None:
            out = torch.zeros((self.npix,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
        for r in range(self.g):
            ids = self.pixel_ids_on(r, first.device)
            out[ids] = g[r][: ids.numel()]
        return out.numpy() if is_np else out


class Gatherer:
    """The exchange step of a frame: ONE gather (RCCL over xGMI on GPUs, gloo on CPU) of every rank's padded tile buffer
    to rank 0, then the un-permute into image order.  Send buffer, receive list, pixel-id tensors and the image are
    allocated once and reused for every frame."""

    def __init__(self, plan: RowTilePlan, rank: int, group=None):
        self.plan, self.rank, self.group = plan, rank, group
        self._buf = self._parts = self._image = None

    def __call__(self, local):
        """local: torch tensor [n_rays(rank), C].  Returns the assembled [H*W, C] image on rank 0, None elsewhere."""
        import torch
        import torch.distributed as dist
        plan = self.plan
        shape = (plan.max_rays(),) + tuple(local.shape[1:])
        if self._buf is None or self._buf.shape != shape or self._buf.dtype != local.dtype or self._buf.device != local.device:
            self._buf = torch.zeros(shape, dtype=local.dtype, device=local.device)
            self._parts = [torch.empty_like(self._buf) for _ in range(plan.g)] if self.rank == 0 else None
            self._image = torch.zeros((plan.npix,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device) if self.rank == 0 else None
        self._buf[: local.shape[0]] = local
        if plan.g == 1 or not dist.is_initialized():
            return plan.assemble([self._buf], out=self._image)
        if self.rank == 0:
            dist.gather(self._buf, gather_list=self._parts, dst=0, group=self.group)
            return plan.assemble(self._parts, out=self._image)
        dist.gather(self._buf, gather_list=None, dst=0, group=self.group)
        return None


def gather_to_root(local, plan: RowTilePlan, rank: int, group=None):
    """One-off form of Gatherer (allocates per call)."""
    return Gatherer(plan, rank, group)(local)


class ShardedRenderer:
    """One rank's share of a frame: holds the rank's rays in HBM and renders its tiles."""

    def __init__(self, ctx, plan: RowTilePlan, rank: int, rays_np: np.ndarray, device):
        import torch
        self.ctx, self.plan, self.rank = ctx, plan, rank
        ids = plan.pixel_ids(rank)
        self.n = int(ids.size)
        self.d_rays = torch.from_numpy(np.ascontiguousarray(rays_np.reshape(-1, 6)[ids])).to(device)
        self.d_rgba = torch.zeros((max(self.n, 1), 4), dtype=torch.uint8, device=device)
        self.d_accum = None

    def render(self, n_samples, seed=1, mode=1, flags=0, want_accum=False, stream=0):
        import torch
        if self.n == 0:
            return self.d_rgba[:0]
        acc_ptr = 0
        if want_accum:
            if self.d_accum is None:
                self.d_accum = torch.zeros((self.n, 3), dtype=torch.float32, device=self.d_rgba.device)
            acc_ptr = self.d_accum.data_ptr()
        self.ctx.render_device(self.d_rays.data_ptr(), self.n, n_samples, self.d_rgba.data_ptr(), seed=seed, mode=mode,
                               flags=flags, shard=self.plan.shard(self.rank), image_width=self.plan.w,
                               d_out_accum=acc_ptr, stream=stream)
        return self.d_rgba[: self.n]
