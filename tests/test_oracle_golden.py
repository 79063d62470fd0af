"""Leg A of the parity chain: the C restatement reproduces the reference's images bit for bit.
Fixtures: tests/golden/golden.json + ref_small_images.npz (outputs of the compiled reference)."""
import numpy as np
import pytest

from spath_amd import view


def _moves(e):
    return [(k, tuple(v) if isinstance(v, list) else v) for k, v in e["moves"]]


def _rays(O, e):
    return O.viewport(e["w"], e["h"], _moves(e))


def test_every_golden_render(O, golden, scenes):
    for e in golden["renders"]:
        if e["w"] * e["h"] * e["spp"] > 2_000_000:
            continue                      # the 1280x720x64 case has its own test below
        t, m = scenes[e["scene"]]
        rays = _rays(O, e)
        if e["mode"] == "flat":
            img = O.render_flat(rays, e["w"], e["h"], t, m)
        else:
            img = O.render_mt(rays, e["w"], e["h"], t, m, e["spp"], e["threads"])
        assert O.fnv1a64(img.tobytes()) == e["fnv1a64"], e
        assert [int(img[:, c].astype(np.int64).sum()) for c in range(3)] == e["sum_rgb"]


def test_survey_hashes(golden):
    """The fixtures reproduce SURVEY.md Appendix B.3 (independently obtained during the survey)."""
    want = {("render", 320, 240, 4, 8): "20cfbe51ee4b43a4", ("render", 320, 240, 4, 1): "571977527ebd73ac",
            ("render", 320, 240, 4, 2): "754440cbabed5f64", ("render", 320, 240, 4, 3): "92a9aeef91dceb74",
            ("render", 320, 240, 4, 64): "1054fb7ff03cf53a", ("flat", 320, 240, 1, 8): "cdfb3998fb314fe9",
            ("render", 1280, 720, 64, 8): "5737eb88c54acec2"}
    seen = 0
    for e in golden["renders"]:
        k = (e["mode"], e["w"], e["h"], e["spp"], e["threads"])
        if e["scene"] == "default" and not e["moves"] and k in want:
            assert e["fnv1a64"] == want[k]
            seen += 1
    assert seen == len(want)


def test_raw_reference_images(O, ref_images, scenes):
    t, m = scenes["default"]
    img = O.render_mt(view.Camera(70, 50).get_viewport(), 70, 50, t, m, 8, 8)
    assert np.array_equal(img, ref_images["default_render_70x50_s8_T8_still"])
    img = O.render_mt(view.Camera(70, 50).get_viewport(), 70, 50, t, m, 8, 3)
    assert np.array_equal(img, ref_images["default_render_70x50_s8_T3_still"])
    moves = [("mov", (0.3, 0.1, -0.5)), ("rot", (0.1, -0.25, 0.0)), ("focal", 0.5)]
    rays = O.viewport(64, 48, moves)
    assert np.array_equal(O.render_flat(rays, 64, 48, t, m), ref_images["default_flat_64x48_s1_T8_moved"])
    assert np.array_equal(O.render_mt(rays, 64, 48, t, m, 4, 8), ref_images["default_render_64x48_s4_T8_moved"])


def test_config2_default_720p_64spp(O, golden, scenes):
    """BASELINE.json configs[1] on the oracle: 1280x720, 64 spp, T=8 (about 3 s of CPU)."""
    e = [x for x in golden["renders"] if (x["w"], x["spp"]) == (1280, 64)][0]
    t, m = scenes["default"]
    img = O.render_mt(O.viewport(1280, 720), 1280, 720, t, m, 64, 8)
    assert [int(img[:, c].astype(np.int64).sum()) for c in range(3)] == e["sum_rgb"]
    assert O.fnv1a64(img.tobytes()) == e["fnv1a64"]


def test_viewports(O, golden):
    for e in golden["viewports"]:
        rays = O.viewport(e["w"], e["h"], _moves(e))
        u = rays.view(np.uint32)
        assert [int(x) for x in u[0]] == e["first_ray_bits"] and [int(x) for x in u[-1]] == e["last_ray_bits"]
        assert [int(np.bitwise_xor.reduce(u[:, c])) for c in range(6)] == e["xor_bits"]
        assert [int(u[:, c].astype(np.uint64).sum() & 0xFFFFFFFFFFFFFFFF) for c in range(6)] == e["sum_bits"]
        if e["fnv1a64"]:
            assert O.fnv1a64(rays.tobytes()) == e["fnv1a64"]


def test_survey_viewport_kat(O):
    """camera(320,240) ray[0] and the centre ray."""
    rays = O.viewport(320, 240).view(np.uint32)
    assert [f"{x:08x}" for x in rays[0]] == ["3f2a2222", "3efeeeef", "c0400000", "3e9d209c", "3e6b71dd", "3f6c6e0f"]
    assert [f"{x:08x}" for x in rays[160 + 120 * 320][3:]] == ["ba88887f", "ba88887f", "3f7fffee"]


def test_counter_rng_variant_is_partition_independent(O, scenes):
    """The counter-RNG oracle (CPU twin of the HIP kernel) gives the same pixels for any pixel range/worker count."""
    t, m = scenes["open_clutter_100"]
    rays = view.Camera(40, 30).get_viewport()
    full, facc, scans = O.render_counter(rays, t, m, 3, seed=42, workers=1)
    a, aacc, s1 = O.render_counter(rays, t, m, 3, seed=42, pix0=0, npix=517, workers=3)
    b, bacc, s2 = O.render_counter(rays, t, m, 3, seed=42, pix0=517, npix=1200 - 517, workers=5)
    assert np.array_equal(np.concatenate([a, b]), full) and np.array_equal(np.concatenate([aacc, bacc]), facc)
    assert s1 + s2 == scans
    other, _, _ = O.render_counter(rays, t, m, 3, seed=43)
    assert not np.array_equal(other, full)
