"""The matrix-pipe side product (sphip_selftest_device what=6) on chosen and random values against double precision."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spath_amd import capi
ctx = capi.Context(0)
f32, f16 = np.float32, np.float16
def ref(rows):
    out = []
    for q in rows:
        s = 16.0 * float(f16(q[10]))
        for k in range(5):
            th = f16(q[k]); tl = f16(f32(q[k]) - f32(th)); rh = f16(q[5 + k]); rl = f16(f32(q[5 + k]) - f32(rh))
            s += float(th) * float(rh) + float(tl) * float(rh) + float(th) * float(rl)
        out.append(s)
    return np.array(out)
cases = np.array([
 [8.594257, 7.673834, 0.9077976, -1.3686118, -0.3599973,  -0.24978559, -0.053257123, 2.8572106, 17.382812, 9.109534, float(f16(1.6895982*0.9999141)), 0],
 [-15.003093, -13.390737, -2.6171205, -3.3570113, -1.2442108,  -0.1808762, -0.054040153, -14.996094, 7.709597, 17.144651, float(f16(0.29708135*1.0001272)), 0],
 [-10.29436, -1.0759379, 0.8339609, 1.1424425, 1.4709595,  0.57389826, 0.6570082, -0.5202796, -16.023438, 14.215171, float(f16(0.27601242*1.0004008)), 0]], dtype=np.float32)
got = ctx.selftest(6, cases, len(cases)).reshape(-1, 2)
print("cases: mfma, device double sum, host reference:"); print(np.c_[got, ref(cases)])
rng = np.random.default_rng(3)
n = 200000
q = np.zeros((n, 12), dtype=np.float32)
q[:, :5] = rng.uniform(-16, 16, (n, 5)); q[:, 5:10] = rng.uniform(-32, 32, (n, 5)) * 10.0 ** rng.uniform(-3, 0, (n, 5)); q[:, 10] = f16(rng.uniform(-30, 30, n)).astype(np.float32)
got = ctx.selftest(6, q, n).reshape(-1, 2)
r = ref(q[:2000])
mag = 16 * np.abs(q[:, 10]) + (np.abs(q[:, :5]) * np.abs(q[:, 5:10])).sum(1)
err = np.abs(got[:, 0].astype(np.float64) - got[:, 1].astype(np.float64)) / mag
print("random: max |mfma - double sum| / sum|terms| =", err.max(), "= 2^%.1f" % np.log2(err.max()), "; host vs device double sum:", np.abs(r - got[:2000, 1]).max())
