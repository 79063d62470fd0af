"""BASELINE.json configs[1] (default 7-triangle scene, 1280x720, 64 spp): time per frame by kernel variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spath_amd import capi, scene, view
ctx = capi.Context(0)
t, m = scene.default_scene(); ctx.set_scene(t, m)
w, h, spp = 1280, 720, 64
rays = view.Camera(w, h).get_viewport()
for var in (0, 1, 2, 6, 12):
    for rep in range(2):
        ctx.render(rays, w, h, spp, flags=var); st = ctx.stats()
    print(f"variant {var}->{st['kernel_variant']}: {st['kernel_ms']:.2f} ms, scans {st['scans_executed']}, {w*h*spp*5/st['kernel_ms']/1e3:.0f} Mray/s nominal, {st['scans_executed']/st['kernel_ms']/1e3:.0f} M scans/s")
