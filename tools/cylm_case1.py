import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spath_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = capi.Context(0)
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "cases", sys.argv[2]))
t, m, ray, want = z["t"], z["m"], z["ray"], int(z["want"])
ctx.set_scene(np.ascontiguousarray(t[want:want + 1]), np.ascontiguousarray(m[want:want + 1]))
rr = np.repeat(ray[None], 64, axis=0).astype(np.float32).copy(); ss = np.full(64, -1, dtype=np.int32)
d_r, d_s = torch.from_numpy(rr).cuda(), torch.from_numpy(ss).cuda()
oi = torch.zeros(64, dtype=torch.int32, device="cuda"); od = torch.zeros(64, dtype=torch.float32, device="cuda")
ctx.closest_hit_device(d_r.data_ptr(), 64, oi.data_ptr(), od.data_ptr(), d_src_idx=d_s.data_ptr(), flags=16); torch.cuda.synchronize()
print("result", int(oi[0]), float(od[0]), "tri", t[want, :9], "ray", ray)
