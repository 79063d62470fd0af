"""spath_amd -- MI355X (gfx950) path-tracing backend for Emanem/spath.

    capi      ctypes binding of the C ABI (include/spath_hip.h -> libspath_hip.so); no CPU fallback
    renderer  Python mirror of the reference's renderer plugin interface (scene::renderer, basic_renderer, get(w,h))
    view      camera / viewport (view::camera::get_viewport) in bit-exact float32
    scene     triangle / material arrays: the reference's default scene, synthetic closed-room scenes, scene files
    dist      pixel-row-tile sharding across GPUs and the gather that reassembles the frame
    csrc/     HIP kernels and the C ABI implementation        host/   C++ adapter (hip_renderer) and headless CLI
"""

__all__ = ["capi", "renderer", "view", "scene", "dist"]
