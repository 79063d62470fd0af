"""Timing experiments on the standalone closest-hit kernel (fixed number of scans whatever the results)."""
import os, sys, glob, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import numpy as np, torch
    from spath_amd import capi
    capi.LIB_PATH = sys.argv[1]
    from spath_amd import scene
    ctx = capi.Context(0)
    ntri = int(os.environ.get("NTRI", "10000"))
    t, m = scene.closed_room(ntri); ctx.set_scene(t, m)
    n = 1920 * 1080 * 4
    rng = np.random.default_rng(1)
    rays = np.concatenate([rng.uniform(-1.4, 1.4, (n, 3)) * [1, 0.5, 1], rng.normal(size=(n, 3))], axis=1).astype(np.float32)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    d_r = torch.from_numpy(rays).cuda(); d_i = torch.zeros(n, dtype=torch.int32, device="cuda"); d_d = torch.zeros(n, dtype=torch.float32, device="cuda")
    for var in [int(x) for x in os.environ.get("VARS", "3").split(",")]:
        for rep in range(2):
            ctx.closest_hit_device(d_r.data_ptr(), n, d_i.data_ptr(), d_d.data_ptr(), flags=var); st = ctx.stats()
        print(f"{os.path.basename(sys.argv[1]):50s} var={var} {st['kernel_ms']:8.1f} ms  {n*ntri/st['kernel_ms']/1e9:.3f} T tests/s  hits={(d_i>=0).float().mean().item():.3f}", flush=True)
else:
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for lib in sorted(glob.glob(os.path.join(root, "build", "abl_*.so"))):
        subprocess.run([sys.executable, __file__, lib])
