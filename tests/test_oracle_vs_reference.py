"""Live leg A: C restatement vs the compiled reference on configurations not stored as fixtures.
Runs wherever oracle/_ref/spath_ref exists (build container; the binary also travels to the GPU box)."""
import numpy as np
import pytest

from spath_amd import scene


@pytest.fixture(scope="module")
def ref(O):
    if not O.have_ref():
        pytest.skip("oracle/_ref/spath_ref not built (no /root/reference here)")
    return O


@pytest.mark.parametrize("w,h,spp,T", [(50, 37, 3, 8), (50, 37, 3, 5), (16, 16, 6, 16), (81, 3, 2, 7), (5, 3, 2, 8)])
def test_default_scene_shapes_and_threads(ref, scenes, w, h, spp, T):
    t, m = scenes["default"]
    rays = ref.viewport(w, h)
    assert np.array_equal(ref.ref_run("viewport", w, h), rays)
    assert np.array_equal(ref.ref_run("render", w, h, spp, t, m, threads=T), ref.render_mt(rays, w, h, t, m, spp, T))
    assert np.array_equal(ref.ref_run("flat", w, h, 1, t, m), ref.render_flat(rays, w, h, t, m))


def test_synthetic_scenes_and_arbitrary_rays(ref):
    rng = np.random.default_rng(3)
    for t, m in (scene.closed_room(64, seed=11), scene.open_clutter(40, seed=5)):
        w, h = 24, 18
        # rays that are not a camera viewport: random origins inside the room, random (unnormalised) directions
        rays = np.concatenate([rng.uniform(-1.0, 1.0, (w * h, 3)), rng.normal(size=(w * h, 3))], axis=1).astype(np.float32)
        a = ref.ref_run("render", w, h, 2, t, m, threads=8, rays=rays)
        b = ref.render_mt(rays, w, h, t, m, 2, 8)
        assert np.array_equal(a, b)
        assert np.array_equal(ref.ref_run("flat", w, h, 1, t, m, rays=rays), ref.render_flat(rays, w, h, t, m))


def test_degenerate_triangles(ref):
    """Zero-area and duplicate triangles, NaN normal of a degenerate triangle never reached."""
    t, m = scene.default_scene()
    t = np.concatenate([t, t[:2]]).copy()
    m = np.concatenate([m, m[:2]]).copy()
    t[7, 3:9] = t[7, 0:3].tolist() * 2          # all three vertices equal
    w, h = 40, 30
    rays = ref.viewport(w, h)
    assert np.array_equal(ref.ref_run("render", w, h, 3, t, m, threads=8), ref.render_mt(rays, w, h, t, m, 3, 8))
    assert np.array_equal(ref.ref_run("flat", w, h, 1, t, m), ref.render_flat(rays, w, h, t, m))
