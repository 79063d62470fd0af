import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spath_amd import capi
capi.LIB_PATH = sys.argv[1]
from spath_amd import scene, view
ctx = capi.Context(0)
t, m = scene.closed_room(10000); ctx.set_scene(t, m)
w, h, spp = 960, 540, 4
rays = view.Camera(w, h).get_viewport()
for var in (6,):
    ctx.render(rays, w, h, spp, flags=var); st = ctx.stats()
    print(var, st["kernel_ms"], st["scans_executed"])
