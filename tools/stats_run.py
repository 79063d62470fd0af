import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spath_amd import capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "..", "build", "libspath_hip_stats.so")
from spath_amd import scene, view
ctx = capi.Context(0)
t, m = scene.closed_room(10000); ctx.set_scene(t, m)
w, h, spp = 960, 540, 2
rays = view.Camera(w, h).get_viewport()
for var in (3,):
    ctx.render(rays, w, h, spp, flags=var); st = ctx.stats()
    print(var, st["kernel_ms"], st["scans_executed"])
