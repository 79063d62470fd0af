#!/usr/bin/env python3
"""Regenerate tests/golden/ from the compiled, unmodified reference (oracle/_ref/spath_ref).

Run in the build container (needs /root/reference to build oracle/_ref):
    python tests/golden/make_golden.py

Writes
  golden.json          -- known-answer lines of the reference's inline functions (seed_dist, rand_unit_vec,
                          std::sin/cos hashes over the LCG's angles, flat_normal, ray_intersect, vec3_RGBA,
                          constants) and FNV-1a-64 hashes / channel sums of reference renders and viewports
  ref_default_64x48.npz -- small raw reference images (RGBA8) for byte-level comparison

Everything in these files is OUTPUT DATA of the reference binary; no reference source text is stored.
The scenes fed to it come from spath_amd/scene.py (values of the default scene: SURVEY.md Appendix C).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O          # noqa: E402
from spath_amd import scene             # noqa: E402


def img_entry(img):
    return {"fnv1a64": O.fnv1a64(img.tobytes()), "sum_rgb": [int(img[:, c].astype(np.int64).sum()) for c in range(3)],
            "nonblack": int((img[:, :3].astype(np.int64).sum(axis=1) > 0).sum())}


def main():
    O.build()
    assert O.have_ref(), "oracle/_ref/spath_ref is required (build container only)"
    G = {"generator": "tests/golden/make_golden.py", "source": "oracle/_ref/spath_ref (unmodified reference cpu_renderer.cpp, "
         "g++ -std=c++11 -O3 -D_RELEASE -pthread -Wno-narrowing -ffp-contract=off)", "kat": O.ref_kat().splitlines(),
         "renders": [], "viewports": []}
    dt, dm = scene.default_scene()
    scenes = {
        "default": (dt, dm),
        "closed_room_200": scene.closed_room(200),
        "open_clutter_100": scene.open_clutter(100),
    }
    cases = [
        # scene, mode, w, h, spp, T, camera moves
        ("default", "flat", 320, 240, 1, 8, ()),
        ("default", "render", 320, 240, 4, 8, ()),
        ("default", "render", 320, 240, 4, 1, ()),
        ("default", "render", 320, 240, 4, 2, ()),
        ("default", "render", 320, 240, 4, 3, ()),
        ("default", "render", 320, 240, 4, 64, ()),
        ("default", "render", 70, 50, 8, 8, ()),         # 218 chunks, 27 per thread, 44-pixel remainder path
        ("default", "render", 70, 50, 8, 3, ()),
        ("default", "render", 33, 7, 5, 8, ()),          # fewer chunks than threads: everything in the remainder
        ("default", "render", 64, 48, 4, 8, (("mov", (0.3, 0.1, -0.5)), ("rot", (0.1, -0.25, 0.0)), ("focal", 0.5))),
        ("default", "flat", 64, 48, 1, 8, (("mov", (0.3, 0.1, -0.5)), ("rot", (0.1, -0.25, 0.0)), ("focal", 0.5))),
        ("closed_room_200", "render", 48, 32, 2, 8, ()),
        ("closed_room_200", "flat", 96, 64, 1, 8, ()),
        ("open_clutter_100", "render", 48, 32, 3, 8, ()),
        ("open_clutter_100", "flat", 96, 64, 1, 8, ()),
        ("default", "render", 1280, 720, 64, 8, ()),     # BASELINE.json configs[1]
    ]
    raw = {}
    for name, mode, w, h, spp, T, moves in cases:
        t, m = scenes[name]
        img = O.ref_run(mode, w, h, spp, t, m, threads=T, moves=moves)
        e = {"scene": name, "mode": mode, "w": w, "h": h, "spp": spp, "threads": T, "moves": [list(x) for x in moves]}
        e.update(img_entry(img))
        G["renders"].append(e)
        print(e)
        if (w, h) == (64, 48) or (name, w, h, spp) == ("default", 70, 50, 8):
            raw[f"{name}_{mode}_{w}x{h}_s{spp}_T{T}_{'moved' if moves else 'still'}"] = img
    for w, h, moves in [(320, 240, ()), (64, 48, ()), (7, 5, ()), (1920, 1080, ()),
                        (64, 48, (("mov", (0.3, 0.1, -0.5)), ("rot", (0.1, -0.25, 0.0)), ("focal", 0.5))),
                        (40, 30, (("rot", (0.0, 1.0, 0.0)), ("mov", (0.0, 0.0, 1.0))))]:
        rays = O.ref_run("viewport", w, h, moves=moves)
        e = {"w": w, "h": h, "moves": [list(x) for x in moves], "fnv1a64": O.fnv1a64(rays.tobytes()) if w * h <= 80000 else None,
             "first_ray_bits": [int(x) for x in rays.view(np.uint32)[0]], "last_ray_bits": [int(x) for x in rays.view(np.uint32)[-1]],
             "xor_bits": [int(np.bitwise_xor.reduce(rays.view(np.uint32)[:, c])) for c in range(6)],
             "sum_bits": [int(rays.view(np.uint32)[:, c].astype(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF) for c in range(6)]}
        G["viewports"].append(e)
    json.dump(G, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "ref_small_images.npz"), **raw)
    print("wrote golden.json and ref_small_images.npz:", sorted(raw))


if __name__ == "__main__":
    main()
