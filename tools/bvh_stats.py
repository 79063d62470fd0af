import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi
capi.LIB_PATH = sys.argv[1]
from spath_amd import scene, view
ctx = capi.Context(0)
for n in (300, 10000):
    t, m = scene.closed_room(n); ctx.set_scene(t, m)
    rng = np.random.default_rng(1)
    nr = 1 << 18
    o = rng.uniform(-1.4, 1.4, (nr, 3)) * [1, 0.5, 1]; d = rng.normal(size=(nr, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    d_r = torch.from_numpy(rays).cuda(); di = torch.zeros(nr, dtype=torch.int32, device="cuda"); dd = torch.zeros(nr, dtype=torch.float32, device="cuda")
    for rep in range(2):
        ctx.closest_hit_device(d_r.data_ptr(), nr, di.data_ptr(), dd.data_ptr(), flags=capi.FLAG_ACCEL); st = ctx.stats()
    print(n, "tris:", st["kernel_ms"], "ms for", nr, "rays ->", nr / st["kernel_ms"] / 1e3, "M scans/s", flush=True)
