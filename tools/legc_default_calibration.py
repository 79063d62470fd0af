"""Leg C calibration on the default scene (320x240, 1024 spp): how far apart are two renders of the REFERENCE's own integrator
with different RNG streams (thread counts T), and how far is the HIP image (counter RNG, several seeds) from them?
Statistics: image-mean relative difference, 8x8 block-mean L-infinity and mean |difference| (of 255)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from spath_amd import capi, scene, view

w, h, spp = 320, 240, 1024
t, m = scene.default_scene()
rays = view.Camera(w, h).get_viewport()


def bm(img, b=8):
    x = img.reshape(h, w, 4)[: h // b * b, : w // b * b, :3].astype(np.float64)
    return x.reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))


def stats(a, b):
    ma, mb = a[:, :3].astype(np.float64).mean(), b[:, :3].astype(np.float64).mean()
    d = np.abs(bm(a) - bm(b))
    return abs(ma - mb) / mb * 100, d.max(), d.mean()


refs = {T: O.render_mt(rays, w, h, t, m, spp, T) for T in (2, 3, 5, 8, 16, 64)}
print("reference stream vs reference stream (T = simulated host threads of cpu_renderer.cpp:124-170)")
rr = []
for a, b in itertools.combinations(sorted(refs), 2):
    s = stats(refs[a], refs[b]); rr.append(s)
    print(f"  T={a:2d} vs T={b:2d}: image mean {s[0]:.4f} %  block-mean Linf {s[1]:.2f}  mean {s[2]:.3f}")
rr = np.array(rr)
print(f"  -> over {len(rr)} pairs: image mean max {rr[:,0].max():.4f} %, block Linf min/median/max {rr[:,1].min():.2f}/{np.median(rr[:,1]):.2f}/{rr[:,1].max():.2f}, block mean max {rr[:,2].max():.3f}")
ctx = capi.Context(0)
ctx.set_scene(t, m)
print("HIP (counter RNG, seed s) vs reference stream T")
hr = []
for seed in (1, 2, 3, 4, 5, 6):
    img = ctx.render(rays, w, h, spp, seed=seed)
    for T in (8, 64):
        s = stats(img, refs[T]); hr.append(s)
        print(f"  seed {seed} vs T={T:2d}: image mean {s[0]:.4f} %  block-mean Linf {s[1]:.2f}  mean {s[2]:.3f}")
hr = np.array(hr)
print(f"  -> over {len(hr)} pairs: image mean max {hr[:,0].max():.4f} %, block Linf min/median/max {hr[:,1].min():.2f}/{np.median(hr[:,1]):.2f}/{hr[:,1].max():.2f}, block mean max {hr[:,2].max():.3f}")
