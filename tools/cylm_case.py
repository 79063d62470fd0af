"""Replay the saved rpl_cylm mismatch cases (gpurun_out/cylm_case*.npz) with variations of the scene."""
import os, sys, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi
if len(sys.argv) > 1: capi.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = capi.Context(0)
def hit(t, m, ray, src, var):
    ctx.set_scene(np.ascontiguousarray(t), np.ascontiguousarray(m))
    rr = np.repeat(ray[None], 64, axis=0).astype(np.float32).copy(); ss = np.full(64, src, dtype=np.int32)
    d_r, d_s = torch.from_numpy(rr).cuda(), torch.from_numpy(ss).cuda()
    oi = torch.zeros(64, dtype=torch.int32, device="cuda"); od = torch.zeros(64, dtype=torch.float32, device="cuda")
    ctx.closest_hit_device(d_r.data_ptr(), 64, oi.data_ptr(), od.data_ptr(), d_src_idx=d_s.data_ptr(), flags=var); torch.cuda.synchronize()
    return int(oi[0]), float(od[0])
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "cases", "cylm_case*.npz"))):
    z = np.load(f); t, m, ray, src, want = z["t"], z["m"], z["ray"], int(z["src"]), int(z["want"])
    n = t.shape[0]
    print(os.path.basename(f), "n", n, "want", want, "exact:", hit(t, m, ray, src, 2), "cylm:", hit(t, m, ray, src, 16), "cylw4:", hit(t, m, ray, src, 14), flush=True)
    print("   only the target triangle:", hit(t[want:want + 1], m[want:want + 1], ray, -1, 16), " target first + rest:", hit(np.concatenate([t[want:want + 1], t]), np.concatenate([m[want:want + 1], m]), ray, -1, 16), flush=True)
    print("   scene x10:", hit(np.concatenate([t] * 10), np.concatenate([m] * 10), ray, src, 16), " scene x40:", hit(np.concatenate([t] * 40), np.concatenate([m] * 40), ray, src, 16), flush=True)
    for k in range(n):
        keep = [i for i in range(n) if i != k]
        if k == want: continue
        r = hit(t[keep], m[keep], ray, -1, 16)
        print(f"   without triangle {k}: {r}", flush=True)
