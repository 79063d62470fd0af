"""Every BASELINE.json config at its own triangle count, image width and launch shape, on a real MI355X through the C ABI.

configs[2] is rendered whole at its real 256 spp (the sample-chunked launch + the in-order resolve pass that bench.py
times); configs[3] and configs[4] are far too large for a test (4e15 and 1.7e17 ray-triangle pairs), so a declared
slice of each is rendered -- 64 interleaved rows of the 8-GPU row-tile plan of the 3840x2160 frame, at reduced spp --
exactly the way a rank renders its shard (sphip_shard with global pixel keys).  Each case checks
  * the default two-stage scan == the exact-only scan (every pair through geom::ray_intersect, cpu_renderer.cpp:39-49),
    bit for bit: RGBA8, float accumulators, scan counts;
  * the CPU oracle on a band of >= 256 (64 at 1M triangles) consecutive pixels with their global pixel keys.
STATED TOLERANCE: L-infinity = 0 on float accumulators and RGBA8.
"""
import numpy as np
import pytest
import torch

from spath_amd import capi, scene, view
from spath_amd.dist import RowTilePlan

pytestmark = pytest.mark.gpu

EXACT = 2        # rpl_lds: exact-only scan


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _render_shard(hip, d_rays, n, shard, width, spp, seed, flags):
    out = torch.zeros(n, 4, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(n, 3, dtype=torch.float32, device="cuda")
    hip.render_device(d_rays.data_ptr(), n, spp, out.data_ptr(), seed=seed, flags=flags, shard=shard, image_width=width,
                      d_out_accum=acc.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy(), acc.cpu().numpy(), hip.stats()


@pytest.mark.parametrize("n_tris,spp,band_px", [(100000, 8, 256), (1000000, 2, 64)], ids=["configs3_100k_4K", "configs4_1M_4K"])
def test_4k_row_tile_slice_exact_scan_and_oracle(hip, O, n_tris, spp, band_px):
    """configs[3] (100k triangles) / configs[4] (1M triangles), 3840x2160: the first 64 rows (8 tiles of 8 rows, i.e. rows
    0-7, 64-71, ... of the frame) of rank 0's shard in the 8-GPU round-robin row-tile plan."""
    w, h, world, tile_rows, seed = 3840, 2160, 8, 8, 1
    t, m = scene.closed_room(n_tris)
    rays = view.Camera(w, h).get_viewport()
    plan = RowTilePlan(w, h, world, tile_rows)
    ids = plan.pixel_ids(0)[: 8 * plan.tile_px]                 # 64 rows = 245 760 pixels
    n = int(ids.size)
    d_rays = _dev(rays.reshape(-1, 6)[ids])
    d_t, d_m = _dev(t), _dev(m)
    hip.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), n_tris, torch.cuda.current_stream().cuda_stream)
    shard = plan.shard(0)
    a_img, a_acc, a_st = _render_shard(hip, d_rays, n, shard, w, spp, seed, 0)
    assert a_st["kernel_variant"] == 16, a_st        # the library's own choice: the cylinder filter on the f16 matrix pipe
    b_img, b_acc, b_st = _render_shard(hip, d_rays, n, shard, w, spp, seed, EXACT)
    assert b_st["kernel_variant"] == EXACT
    assert np.array_equal(a_img, b_img) and np.array_equal(a_acc, b_acc)
    assert a_st["scans_executed"] == b_st["scans_executed"]
    assert abs(a_st["scans_executed"] - n * spp * 5) <= 2e-4 * n * spp * 5      # closed room: nominal == executed up to edge leaks
    # the other filter scans of the loaded build as well (shipped build: rpl_cylw4s, the f32 cylinder scan; -DSP_ALL_VARIANTS builds:
    # the slab-filter scan and a per-lane cylinder scan too)
    have = set(capi.available_variants())
    for var in [v for v in (6, 15, 13 if spp >= 4 else 12) if v in have]:
        c_img, c_acc, c_st = _render_shard(hip, d_rays, n, shard, w, spp, seed, var)
        assert np.array_equal(c_img, b_img) and np.array_equal(c_acc, b_acc) and c_st["scans_executed"] == b_st["scans_executed"], var
    # oracle on a band in the second tile of the slice: frame rows 64.., i.e. global pixel keys far from the local indices
    k0 = plan.tile_px + 3 * w + 1700
    p0 = int(ids[k0])
    assert p0 == 8 * tile_rows * w + 3 * w + 1700
    want_img, want_acc, _ = O.render_counter(rays, t, m, spp, seed, pix0=p0, npix=band_px)
    assert np.array_equal(a_img[k0:k0 + band_px], want_img) and np.array_equal(a_acc[k0:k0 + band_px], want_acc)


def test_configs2_real_launch_shape_256spp(hip, O):
    """configs[2] exactly as bench.py times it: 10k triangles, 1920x1080, 256 spp, the library's own variant and sample
    chunks (pixel x chunk lanes + k_resolve) against the unsplit launch and an oracle band at the full 256 spp."""
    w, h, spp, seed = 1920, 1080, 256, 1
    t, m = scene.closed_room(10000)
    rays = view.Camera(w, h).get_viewport()
    d_rays, d_t, d_m = _dev(rays), _dev(t), _dev(m)
    hip.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), 10000, torch.cuda.current_stream().cuda_stream)
    a_img, a_acc, a_st = _render_shard(hip, d_rays, w * h, None, w, spp, seed, 0)
    assert a_st["n_launches"] == 2 and a_st["kernel_variant"] == 16            # sample-chunked rpl_cylm + resolve
    b_img, b_acc, b_st = _render_shard(hip, d_rays, w * h, None, w, spp, seed, capi.flag_chunks(1))
    assert b_st["n_launches"] == 1
    assert np.array_equal(a_img, b_img) and np.array_equal(a_acc, b_acc) and a_st["scans_executed"] == b_st["scans_executed"]
    assert abs(a_st["scans_executed"] - w * h * spp * 5) <= 1e-5 * w * h * spp * 5
    # SURVEY 8(f3), opt-in: the primary hit of each pixel from a one-scan-per-pixel pre-pass -- same bits, 1/5 fewer scans
    c_img, c_acc, c_st = _render_shard(hip, d_rays, w * h, None, w, spp, seed, capi.FLAG_PRIMARY_REUSE)
    assert c_st["n_launches"] == 3 and c_st["kernel_variant"] == 16
    assert np.array_equal(a_img, c_img) and np.array_equal(a_acc, c_acc)
    assert c_st["scans_executed"] == a_st["scans_executed"] - w * h * (spp - 1)
    p0, n = 540 * w + 700, 256
    want_img, want_acc, _ = O.render_counter(rays, t, m, spp, seed, pix0=p0, npix=n)
    assert np.array_equal(a_img[p0:p0 + n], want_img) and np.array_equal(a_acc[p0:p0 + n], want_acc)
