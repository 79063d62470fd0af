"""All GPUs of a node behind one context (include/spath_hip.h: sphip_create_multi) -- the north_star's "framebuffer shards by
pixel-row tiles across the GPUs of one node, RCCL gather over xGMI", behind the renderer interface (src/renderer.h:31-32).

CPU part: the row-tile plan the library computes in C++ equals the Python plan bench.py uses and deals every pixel once.
GPU part (runs on a one-GPU box): a device may be listed several times, so {0,0,0} renders three shards on one GPU through
the multi-device code path (per-device host threads, tile uploads / per-shard viewport kernels, peer-copy gather into the
first device's buffer, un-permute kernel, one D2H) and must reproduce the single-context image bit for bit; the RCCL
communicator path is exercised with one device.  STATED TOLERANCE: 0."""
import os
import subprocess

import numpy as np
import pytest

from spath_amd import capi, scene, view
from spath_amd.dist import RowTilePlan, balanced_tile_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "spath_amd", "host", "build", "spath_cli")


def test_library_plan_equals_python_plan_and_covers_every_pixel_once():
    for (w, h, g) in [(1920, 1080, 8), (3840, 2160, 8), (1920, 1080, 1), (17, 13, 4), (8, 3, 8), (5, 2, 4), (64, 1081, 8), (33, 20, 3)]:
        tr = capi.plan_tile_rows(h, g)
        assert tr == balanced_tile_rows(h, g)
        plan = RowTilePlan(w, h, g, tr)
        seen = np.zeros(w * h, dtype=np.int32)
        for r in range(g):
            shard, n = capi.plan_shard(w, h, g, tr, r)
            assert shard == plan.shard(r) and n == plan.n_rays(r)
            k = np.arange(n, dtype=np.int64)
            pix = shard[0] + (k // shard[1]) * shard[2] + (k % shard[1])        # the sphip_shard formula
            assert np.array_equal(pix, plan.pixel_ids(r))
            seen[pix] += 1
        assert (seen == 1).all()
    with pytest.raises(capi.SpathHipError):
        capi.plan_shard(10, 10, 4, 2, 4)                                       # rank out of range
    assert capi.plan_tile_rows(0, 4) < 0


def test_multi_context_without_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.SpathHipError, match="sphip_create_multi"):
        capi.Context.multi([0, 0])
    with pytest.raises(capi.SpathHipError):
        capi.Context.multi()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 8])
def test_multi_device_context_reproduces_the_single_device_image(hip, O, devices):
    t, m = scene.open_clutter(300)
    hip.set_scene(t, m)
    mc = capi.Context.multi(devices)
    assert mc.device_count == len(devices)
    assert "HIP - Path Tracing" in mc.description
    mc.set_scene(t, m)
    for (w, h, spp) in [(96, 70, 3), (61, 37, 2), (128, 64, 1), (16, 5, 2)]:        # fewer tiles than devices at the end
        cam = view.Camera(w, h)
        cam.set_delta_mov((0.1, 0.0, -0.2)); cam.set_delta_rot((0.0, 0.15, 0.0))
        rays = cam.get_viewport()
        want_img, want_acc = hip.render(rays, w, h, spp, seed=21, want_accum=True)
        want_scans = hip.stats()["scans_executed"]
        img, acc = mc.render(rays, w, h, spp, seed=21, want_accum=True)
        st = mc.stats()
        assert np.array_equal(img, want_img) and np.array_equal(acc, want_acc), (devices, w, h)
        assert st["scans_executed"] == want_scans
        if len(devices) > 1:
            assert 1 <= st["n_devices"] <= len(devices) and st["gather_kind"] == capi.GATHER_PEER and st["kernel_ms_min"] <= st["kernel_ms"]
        # rays generated on each device for its own tiles (no 24 B/pixel upload)
        img2, acc2 = mc.render_camera(cam, spp, seed=21, want_accum=True)
        assert np.array_equal(img2, want_img) and np.array_equal(acc2, want_acc)
        # flat pass, and the RGBA8-only path
        assert np.array_equal(mc.render(rays, w, h, 1, mode=capi.MODE_FLAT), hip.render(rays, w, h, 1, mode=capi.MODE_FLAT))
        assert np.array_equal(mc.render(rays, w, h, spp, seed=21), want_img)
    assert np.array_equal(mc.render(rays, w, h, spp, seed=21), O.render_counter(rays, t, m, spp, 21)[0])
    if len(devices) > 1:
        with pytest.raises(capi.SpathHipError, match="single-device"):
            mc.render_device(1, 64, 1, 1)
    with pytest.raises(capi.SpathHipError, match="n_samples"):
        mc.render(rays, w, h, 0)
    mc.close()


@pytest.mark.gpu
def test_multi_device_full_frame_and_rccl_communicator(hip):
    """configs[2]'s frame on eight shards of one GPU == the single-context frame; and SPATH_HIP_GATHER=rccl with the one
    device there is: librccl loads, ncclCommInitAll / ncclCommDestroy run (the 8-GPU exchange itself needs 8 GPUs)."""
    t, m = scene.closed_room(10000)
    w, h, spp = 1920, 1080, 2
    rays = view.Camera(w, h).get_viewport()
    hip.set_scene(t, m)
    want = hip.render(rays, w, h, spp, seed=1)
    mc = capi.Context.multi([0] * 8)
    mc.set_scene(t, m)
    assert np.array_equal(mc.render(rays, w, h, spp, seed=1), want)
    st = mc.stats()
    assert st["n_devices"] == 8 and st["scans_executed"] == hip.stats()["scans_executed"]
    mc.close()
    env = dict(os.environ, SPATH_HIP_GATHER="rccl")
    one = ("import numpy as np\nfrom spath_amd import capi, scene, view\n"
           "t, m = scene.closed_room(300); rays = view.Camera(96, 64).get_viewport()\n"
           "a = capi.Context(0); a.set_scene(t, m); want = a.render(rays, 96, 64, 3, seed=5)\n"
           "c = capi.Context.multi([0]); c.set_scene(t, m); got = c.render(rays, 96, 64, 3, seed=5); st = c.stats()\n"
           "print('rccl-one', bool(np.array_equal(got, want)), st['gather_kind'], c.device_count, c.description)\n")
    p = subprocess.run([os.sys.executable, "-c", one], capture_output=True, text=True, env=env, cwd=ROOT)
    assert "rccl-one True 1 1" in p.stdout and "RCCL gather" in p.stdout, p.stdout + p.stderr[-2000:]
    # the same communicator of one on the frame shapes that leave ranks without tiles on a real node ((16, 5): two tiles for eight
    # devices), a single row, a single pixel, the flat pass, the camera path and accumulators: the RCCL branch of multi_render
    # (group start / end with nothing to send, local copy of the root's own tiles, assemble, D2H) as far as one device reaches
    shapes = ("import numpy as np\nfrom spath_amd import capi, scene, view\n"
              "t, m = scene.closed_room(300)\na = capi.Context(0); a.set_scene(t, m)\nc = capi.Context.multi([0]); c.set_scene(t, m)\nok = True\n"
              "for (w, h, spp) in [(16, 5, 3), (97, 1, 2), (1, 1, 4), (33, 20, 1)]:\n"
              "    cam = view.Camera(w, h); rays = cam.get_viewport()\n"
              "    wi, wa = a.render(rays, w, h, spp, seed=9, want_accum=True); gi, ga = c.render(rays, w, h, spp, seed=9, want_accum=True)\n"
              "    ok &= bool(np.array_equal(wi, gi) and np.array_equal(wa, ga)) and c.stats()['gather_kind'] == 1 and c.stats()['scans_executed'] == a.stats()['scans_executed']\n"
              "    ok &= bool(np.array_equal(a.render(rays, w, h, 1, mode=capi.MODE_FLAT), c.render(rays, w, h, 1, mode=capi.MODE_FLAT)))\n"
              "    ok &= bool(np.array_equal(c.render_camera(cam, spp, seed=9), wi))\n"
              "print('rccl-shapes', ok)\n")
    p = subprocess.run([os.sys.executable, "-c", shapes], capture_output=True, text=True, env=env, cwd=ROOT)
    assert "rccl-shapes True" in p.stdout, p.stdout + p.stderr[-2000:]
    p = subprocess.run([os.sys.executable, "-c",
                        "from spath_amd import capi\n"
                        "import ctypes as C\n"
                        "L = capi.load(); h = C.c_void_p(); ids = (C.c_int * 2)(0, 0)\n"
                        "rc = L.sphip_create_multi(ids, 2, C.byref(h)); print('dup', rc, L.sphip_last_error(None).decode())\n"],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    assert "dup -1" in p.stdout and "distinct devices" in p.stdout, p.stdout + p.stderr      # a communicator holds a GPU once


@pytest.mark.gpu
def test_all_visible_devices_when_there_are_several(hip):
    """On a node with >= 2 GPUs: one context over all of them (distinct devices: the RCCL exchange, or peer copies if RCCL does not
    come up) must reproduce device 0's image bit for bit.  A one-GPU box skips this; the several-shards-on-one-GPU tests above cover
    everything but the cross-device transfers themselves."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU on this box")
    t, m = scene.closed_room(2000)
    hip.set_scene(t, m)
    mc = capi.Context.multi(list(range(n)))
    assert mc.device_count == n
    mc.set_scene(t, m)
    for (w, h, spp) in [(320, 180, 4), (97, 61, 5)]:
        cam = view.Camera(w, h)
        rays = cam.get_viewport()
        want_img, want_acc = hip.render(rays, w, h, spp, seed=3, want_accum=True)
        img, acc = mc.render(rays, w, h, spp, seed=3, want_accum=True)
        st = mc.stats()
        assert np.array_equal(img, want_img) and np.array_equal(acc, want_acc)
        assert st["gather_kind"] in (capi.GATHER_RCCL, capi.GATHER_PEER) and st["scans_executed"] == hip.stats()["scans_executed"]
        img2, acc2 = mc.render_camera(cam, spp, seed=3, want_accum=True)
        assert np.array_equal(img2, want_img) and np.array_equal(acc2, want_acc)
    mc.close()


@pytest.mark.gpu
def test_device_list_from_the_environment():
    """sphip_create_multi(NULL): SPATH_HIP_DEVICES picks the devices (what hip_renderer::get relies on); nonsense is a loud error."""
    code = ("from spath_amd import capi\n"
            "c = capi.Context.multi()\n"
            "print('devices', c.device_count, c.description)\n")
    for env_val, want in (("0,0,0", "devices 3"), ("0", "devices 1")):
        p = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SPATH_HIP_DEVICES=env_val), cwd=ROOT)
        assert want in p.stdout, p.stdout + p.stderr[-1500:]
    for bad in ("0;1", "zero", "0,,x", "99"):
        p = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SPATH_HIP_DEVICES=bad), cwd=ROOT)
        assert p.returncode != 0 and "sphip_create_multi" in p.stderr, (bad, p.stdout, p.stderr[-500:])


@pytest.mark.gpu
def test_cli_on_several_shards(tmp_path, O):
    """spath_cli --devices 0,0,0: the C++ adapter (hip_renderer::get_on) over a multi-device context."""
    t, m = scene.default_scene()
    out = os.path.join(tmp_path, "a.rgba")
    w, h = 64, 48
    rays = O.viewport(w, h)
    for extra in (["--devices", "0,0,0"], ["--devices", "0,0,0", "--device-viewport"], ["--gpus", "1"]):
        p = subprocess.run([CLI, "--w", str(w), "--h", str(h), "--spp", "5", "--seed", "77", "--out", out] + extra, check=True, capture_output=True, text=True)
        assert np.array_equal(np.fromfile(out, dtype=np.uint8).reshape(-1, 4), O.render_counter(rays, t, m, 5, 77)[0]), extra
        assert ("[3 device(s)]" if "0,0,0" in extra else "[1 device(s)]") in p.stdout
