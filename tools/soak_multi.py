"""Randomized soak of the multi-device context (several shards on one GPU) against the single-device context: python tools/soak_multi.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spath_amd import capi, scene, view
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(11)
one = capi.Context(0)
ctxs = {k: capi.Context.multi([0] * k) for k in (2, 3, 5, 8)}
t_end, it, fails = time.time() + budget, 0, 0
while time.time() < t_end:
    it += 1
    n = int(rng.integers(14, 3000))
    t, m = scene.closed_room(n, seed=int(rng.integers(1, 1 << 30))) if rng.random() < 0.5 else scene.open_clutter(max(n, 7), seed=int(rng.integers(1, 1 << 30)))
    w, h, spp, seed = int(rng.integers(1, 300)), int(rng.integers(1, 200)), int(rng.choice([1, 2, 4, 5, 9])), int(rng.integers(0, 1 << 60))
    cam = view.Camera(w, h); cam.set_delta_mov(tuple(rng.uniform(-0.4, 0.4, 3))); cam.set_delta_rot(tuple(rng.uniform(-0.3, 0.3, 3)))
    rays = cam.get_viewport()
    one.set_scene(t, m)
    want = one.render(rays, w, h, spp, seed=seed, want_accum=True); ws = one.stats()["scans_executed"]
    wflat = one.render(rays, w, h, 1, mode=capi.MODE_FLAT)
    k = int(rng.choice([2, 3, 5, 8])); mc = ctxs[k]
    mc.set_scene(t, m)
    a = mc.render(rays, w, h, spp, seed=seed, want_accum=True); st = mc.stats()
    b = mc.render_camera(cam, spp, seed=seed, want_accum=True)
    f = mc.render(rays, w, h, 1, mode=capi.MODE_FLAT)
    ok = all(np.array_equal(x, y) for x, y in ((a[0], want[0]), (a[1], want[1]), (b[0], want[0]), (b[1], want[1]), (f, wflat))) and st["scans_executed"] == ws
    if not ok:
        fails += 1; print(f"MISMATCH it {it}: {k} shards, n={n}, {w}x{h}, spp {spp}", flush=True)
    if it % 25 == 0: print(f"  {it} frames, {fails} mismatches", flush=True)
print(f"multi-device soak: {it} random frames on 2/3/5/8 shards (host rays, camera path, flat pass) against the single context: {fails} mismatches", flush=True)
sys.exit(1 if fails else 0)
