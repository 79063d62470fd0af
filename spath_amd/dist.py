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

    def assemble(self, gathered):
        """gathered: [G, max_rays, C] (numpy or torch) -> [H*W, C] image in global pixel order."""
        import torch
        is_np = isinstance(gathered, np.ndarray)
        g = torch.from_numpy(gathered) if is_np else gathered
        out = torch.zeros((self.npix,) + tuple(g.shape[2:]), dtype=g.dtype, device=g.device)
        for r in range(self.g):
            ids = torch.from_numpy(self.pixel_ids(r)).to(g.device)
            out[ids] = g[r, : ids.numel()]
        return out.numpy() if is_np else out


def gather_to_root(local, plan: RowTilePlan, rank: int, group=None):
    """One gather (RCCL over xGMI on GPUs, gloo on CPU) of every rank's padded tile buffer to rank 0.

    local: torch tensor [n_rays(rank), C].  Returns the assembled [H*W, C] image on rank 0, None elsewhere.
    """
    import torch
    import torch.distributed as dist
    pad = plan.max_rays()
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    if plan.g == 1 or not dist.is_initialized():
        return plan.assemble(buf.unsqueeze(0))
    if rank == 0:
        parts = [torch.empty_like(buf) for _ in range(plan.g)]
        dist.gather(buf, gather_list=parts, dst=0, group=group)
        return plan.assemble(torch.stack(parts))
    dist.gather(buf, gather_list=None, dst=0, group=group)
    return None


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
