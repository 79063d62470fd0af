import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from spath_amd import capi, scene, view
ctx = capi.Context(0)
t, m = scene.closed_room(300)
w, h, spp = 96, 64, 1024
rays = view.Camera(w, h).get_viewport()
ctx.set_scene(t, m)
got = ctx.render(rays, w, h, spp, seed=1)
refs = {T: O.render_mt(rays, w, h, t, m, spp, T) for T in (8, 64)}
def bm(img): 
    x = img.reshape(h, w, 4)[:, :, :3].astype(np.float64); return x.reshape(h//8, 8, w//8, 8, 3).mean(axis=(1,3))
for name, a, b in (("gpu vs ref T8", got, refs[8]), ("gpu vs ref T64", got, refs[64]), ("ref T8 vs ref T64", refs[8], refs[64])):
    ma, mb = a[:, :3].astype(float).mean(), b[:, :3].astype(float).mean()
    d = np.abs(bm(a) - bm(b))
    print(name, "mean", ma, mb, "rel", abs(ma-mb)/mb, "block Linf", d.max(), "block mean", d.mean())
