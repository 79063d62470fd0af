"""First contact of a new scan build: every build/abl_*.so renders the configs[2] frame at a few spp (timing, SHA-256 of RGBA8 and
accumulators: all builds must agree with the exact-only scan of the first one) + the large scenes (clutter x 10, 100k, 1M triangles).
python tools/first_contact.py [spp]"""
import glob, hashlib, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
code = r"""
import hashlib, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from spath_amd import capi
capi.LIB_PATH = sys.argv[1]
spp = int(sys.argv[2]); exact = sys.argv[3] == "1"
from spath_amd import scene, view
ctx = capi.Context(0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def run(tag, t, m, rays, w, h, spp, flags, reps=2):
    nt = t.shape[0]
    d_t, d_m, d_r = d(t), d(m), d(rays)
    ctx.set_scene_device(d_t.data_ptr(), d_m.data_ptr(), nt, 0)
    out = torch.zeros(w * h, 4, dtype=torch.uint8, device='cuda'); acc = torch.zeros(w * h, 3, dtype=torch.float32, device='cuda')
    best = 1e30
    for rep in range(reps):
        ctx.render_device(d_r.data_ptr(), w * h, spp, out.data_ptr(), seed=1, flags=flags, d_out_accum=acc.data_ptr()); torch.cuda.synchronize()
        st = ctx.stats(); best = min(best, st['kernel_ms'])
    sha = hashlib.sha256(out.cpu().numpy().tobytes() + acc.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{tag}: flags {flags} variant {st['kernel_variant']} {best:.1f} ms, {w*h*spp*5/best/1e3:.1f} Mray/s, {st['scans_executed']*nt/best/1e9:.3f} T tests/s, scans {st['scans_executed']}, sha {sha}", flush=True)
t, m = scene.closed_room(10000)
rays = view.Camera(1920, 1080).get_viewport()
run("configs[2] 10k 1080p x %%d spp" %% spp, t, m, rays, 1920, 1080, spp, 0)
if exact: run("configs[2] exact-only x 2 spp", t, m, rays, 1920, 1080, 2, 2, reps=1)
run("configs[2] default x 2 spp", t, m, rays, 1920, 1080, 2, 0, reps=1)
t, m = scene.closed_room(10000, clutter_scale=10.0)
run("clutter x10 10k 1080p x 4 spp", t, m, rays, 1920, 1080, 4, 0)
r4 = view.Camera(3840, 2160).get_viewport().reshape(-1, 6)
t, m = scene.closed_room(100000)
run("100k tris, 4K rows 0-269 x 8 spp", t, m, r4[:3840 * 270], 3840, 270, 8, 0)
t, m = scene.closed_room(1000000)
run("1M tris, 4K rows 0-63 x 4 spp", t, m, r4[:3840 * 64], 3840, 64, 4, 0)
""" % root
first = True
for lib in sorted(glob.glob(os.path.join(root, "build", "abl_*.so"))):
    p = subprocess.run([sys.executable, "-c", code, lib, str(spp), "1" if first else "0"], capture_output=True, text=True)
    first = False
    for l in p.stdout.strip().splitlines(): print(f"{os.path.basename(lib):20s} {l}", flush=True)
    if p.returncode: print("FAILED", p.stderr[-1500:], flush=True)
