"""The default scan exists in two workgroup shapes (spath_amd/csrc/sp_cylm_both.h): 256 threads with 256-triangle tiles for scenes
below 16384 triangles, 512 threads with 512-triangle tiles above.  The library picks by scene size, so the small test scenes of the
other files only ever meet the first shape and the 1e5 / 1e6-triangle tests only the second.  Here SPATH_HIP_CYLM_SHAPE forces
each shape onto the same small scenes: closest hits against the oracle (adversarial rays, idx_source), renders against the oracle
(RGBA8, accumulators, scan counts: L-infinity = 0), triangle counts around the tile boundaries of both shapes, and the per-pair
audit of stage 1 (tests/test_hip_stage1_audit.py) under each shape."""
import os

import numpy as np
import pytest
import torch

from spath_amd import capi, scene, view

import test_hip_parity as P
import test_hip_stage1_audit as A

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[256, 512])
def shaped(request):
    old = os.environ.get("SPATH_HIP_CYLM_SHAPE")
    os.environ["SPATH_HIP_CYLM_SHAPE"] = str(request.param)
    ctx = capi.Context(0)
    yield ctx, request.param
    ctx.close()
    if old is None:
        del os.environ["SPATH_HIP_CYLM_SHAPE"]
    else:
        os.environ["SPATH_HIP_CYLM_SHAPE"] = old


def _tiles(ctx, rays):
    """(tiles, triangles per tile) of the stream the context built: which shape really ran."""
    import ctypes as C
    t = C.c_uint32(0)
    ctx._check(ctx._L.sphip_selftest_stage1(ctx._h, None, 0, None, None, None, C.byref(t)), "sphip_selftest_stage1")
    return t.value & 0xFFFFF, (t.value >> 20) & 0x7FF            # (bit 31: octet bits)


def test_forced_shape_closest_hits_and_renders(shaped, O):
    hip, shape = shaped
    rng = np.random.default_rng(31)
    for name, (t, m) in {"closed1k": scene.closed_room(1000), "open300": scene.open_clutter(300), "bigtris": scene.closed_room(600, clutter_scale=10.0),
                         "n255": scene.closed_room(255), "n257": scene.closed_room(257), "n511": scene.closed_room(511), "n513": scene.closed_room(513),
                         "n1537": scene.closed_room(1537)}.items():
        rays = P._adversarial_rays(t, rng)[:6000]
        n = rays.shape[0]
        src = rng.integers(-1, t.shape[0], n).astype(np.int32)
        hip.set_scene(t, m)
        d_rays, d_src = P.dev(rays), P.dev(src)
        d_idx = torch.zeros(n, dtype=torch.int32, device="cuda"); d_d = torch.zeros(n, dtype=torch.float32, device="cuda")
        for use_src in (False, True):
            want_idx, want_d = O.closest_hits(rays, t, src if use_src else None)
            hip.closest_hit_device(d_rays.data_ptr(), n, d_idx.data_ptr(), d_d.data_ptr(), d_src_idx=d_src.data_ptr() if use_src else 0, flags=0)
            torch.cuda.synchronize()
            assert hip.stats()["kernel_variant"] == 16
            assert np.array_equal(d_idx.cpu().numpy(), want_idx), (shape, name, use_src)
            assert np.array_equal(d_d.cpu().numpy().view(np.uint32), want_d.view(np.uint32)), (shape, name, use_src)
        assert _tiles(hip, rays)[1] == shape
        # render_flat and render against the oracle
        w, h, spp, seed = 64, 40, 3, 5
        vr = view.Camera(w, h).get_viewport()
        assert np.array_equal(hip.render(vr, w, h, 1, mode=capi.MODE_FLAT), O.render_flat(vr, w, h, t, m)), (shape, name)
        img, acc = hip.render(vr, w, h, spp, seed=seed, want_accum=True)
        want_img, want_acc, want_scans = O.render_counter(vr, t, m, spp, seed)
        assert np.array_equal(img, want_img) and np.array_equal(acc, want_acc) and hip.stats()["scans_executed"] == want_scans, (shape, name)


def test_forced_shape_stage1_audit(shaped, O):
    hip, shape = shaped
    pairs, acc = A.run_cases(hip, O, fuzz_soups=(1,), fuzz_rays=1024)
    assert pairs >= 10_000_000 and acc >= 20_000


def test_shape_follows_scene_size(hip):
    """Without the override: 256-triangle tiles below 16384 triangles, 512 from there on."""
    assert "SPATH_HIP_CYLM_SHAPE" not in os.environ
    rays = view.Camera(64, 1).get_viewport()
    for n, want in ((10000, 256), (16383, 256), (16384, 512), (50000, 512)):
        t, m = scene.closed_room(n)
        hip.set_scene(t, m)
        assert _tiles(hip, rays)[1] == want, n
