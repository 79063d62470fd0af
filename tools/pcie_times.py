import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spath_amd import capi, scene, view
ctx = capi.Context(0)
t, m = scene.closed_room(10000); ctx.set_scene(t, m)
for (w, h) in ((1920, 1080), (3840, 2160)):
    rays = view.Camera(w, h).get_viewport()
    for rep in range(3):
        ctx.render(rays, w, h, 2); st = ctx.stats()
    print(f"{w}x{h}: H2D rays {rays.nbytes/1e6:.1f} MB {st['upload_ms']:.2f} ms, kernel {st['kernel_ms']:.1f} ms (2 spp), D2H image {w*h*4/1e6:.1f} MB {st['download_ms']:.2f} ms")
