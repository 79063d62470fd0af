"""The C++ host side: hip_renderer (adapter over the C ABI, deriving from basic_renderer) and the headless CLI.

CPU part: the binaries exist, resolve libspath_hip.so, and fail loudly without a GPU (the reference's GPU
peers throw from their constructors and main() prints 'Exception: ...', src/main.cpp:263-267).
GPU part: images produced through the C++ interface equal the oracle's; and, where the build container
produced oracle/_ref/spath_both (reference cpu_renderer.cpp + hip_renderer.cpp compiled against the
REFERENCE's headers behind one scene::renderer* registry), both backends give the same flat image."""
import os
import subprocess

import numpy as np
import pytest

from spath_amd import scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "spath_amd", "host", "build", "spath_cli")
BOTH = os.path.join(ROOT, "oracle", "_ref", "spath_both")


def test_cli_is_built_and_linked():
    assert os.access(CLI, os.X_OK), "run python -c 'import __graft_entry__ as g; g.build()'"
    ldd = subprocess.run(["ldd", CLI], capture_output=True, text=True).stdout
    assert "libspath_hip.so" in ldd and "not found" not in ldd.split("libspath_hip.so")[1].splitlines()[0]


def test_cli_without_gpu_reports_the_exception():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([CLI, "--w", "8", "--h", "8", "--spp", "1"], capture_output=True, text=True)
    assert p.returncode == 1 and "Exception: hip_renderer:" in p.stderr


@pytest.mark.gpu
def test_cli_images_equal_oracle(tmp_path, O):
    t, m = scene.default_scene()
    out = os.path.join(tmp_path, "a.rgba")
    moves = ["--mov", "0.3", "0.1", "-0.5", "--rot", "0.1", "-0.25", "0.0", "--focal", "0.5"]
    omoves = [("mov", (0.3, 0.1, -0.5)), ("rot", (0.1, -0.25, 0.0)), ("focal", 0.5)]
    for w, h in [(64, 48), (37, 29)]:
        rays = O.viewport(w, h, omoves)
        subprocess.run([CLI, "--w", str(w), "--h", str(h), "--mode", "flat", "--out", out] + moves, check=True, capture_output=True)
        assert np.array_equal(np.fromfile(out, dtype=np.uint8).reshape(-1, 4), O.render_flat(rays, w, h, t, m))
        p = subprocess.run([CLI, "--w", str(w), "--h", str(h), "--spp", "5", "--seed", "77", "--out", out] + moves, check=True, capture_output=True, text=True)
        assert "Current renderer: HIP - Path Tracing" in p.stdout and "Done (" in p.stdout
        assert np.array_equal(np.fromfile(out, dtype=np.uint8).reshape(-1, 4), O.render_counter(rays, t, m, 5, 77)[0])
        # viewport generated on the device from the adapter's camera (C++ float std::cos/std::sin for the trig values)
        subprocess.run([CLI, "--w", str(w), "--h", str(h), "--spp", "5", "--seed", "77", "--device-viewport", "--out", out] + moves, check=True, capture_output=True)
        assert np.array_equal(np.fromfile(out, dtype=np.uint8).reshape(-1, 4), O.render_counter(rays, t, m, 5, 77)[0])
    # scene file + PPM output
    sp = os.path.join(tmp_path, "s.bin")
    ts, ms = scene.closed_room(300)
    scene.write_scene(sp, ts, ms)
    ppm = os.path.join(tmp_path, "a.ppm")
    subprocess.run([CLI, "--scene", sp, "--w", "40", "--h", "30", "--spp", "2", "--out", ppm], check=True, capture_output=True)
    raw = open(ppm, "rb").read()
    assert raw.startswith(b"P6\n40 30\n255\n")
    want = O.render_counter(O.viewport(40, 30), ts, ms, 2, 1)[0][:, :3]
    assert np.array_equal(np.frombuffer(raw[len(b"P6\n40 30\n255\n"):], dtype=np.uint8).reshape(-1, 3), want)
    # PNG output (stored deflate blocks), decoded here with zlib: signature, IHDR, CRCs, filter-0 rows
    import struct, zlib
    png = os.path.join(tmp_path, "a.png")
    subprocess.run([CLI, "--scene", sp, "--w", "40", "--h", "30", "--spp", "2", "--out", png], check=True, capture_output=True)
    raw = open(png, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(typ + data) & 0xffffffff)
        chunks.setdefault(typ, b"")
        chunks[typ] += data
        pos += 12 + n
    assert struct.unpack(">IIBBBBB", chunks[b"IHDR"]) == (40, 30, 8, 2, 0, 0, 0) and b"IEND" in chunks
    rows = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), dtype=np.uint8).reshape(30, 1 + 3 * 40)
    assert not rows[:, 0].any() and np.array_equal(rows[:, 1:].reshape(-1, 3), want)


@pytest.mark.gpu
def test_reference_registry_runs_both_backends(tmp_path, O):
    """cpu_renderer (the reference's own object code) and hip_renderer side by side behind scene::renderer*."""
    if not os.access(BOTH, os.X_OK):
        pytest.skip("oracle/_ref/spath_both not built (needs the reference tree at build time)")
    t, m = scene.open_clutter(120)
    sp = os.path.join(tmp_path, "s.bin")
    scene.write_scene(sp, t, m)
    pre = os.path.join(tmp_path, "o")
    env = dict(os.environ, ORACLE_THREADS="8")
    p = subprocess.run([BOTH, sp, "72", "54", "3", pre], check=True, capture_output=True, text=True, env=env)
    assert "CPU - Path Tracing" in p.stdout and "HIP - Path Tracing" in p.stdout
    load = lambda k, mo: np.fromfile(f"{pre}.{k}.{mo}.rgba", dtype=np.uint8).reshape(-1, 4)
    assert np.array_equal(load(0, "flat"), load(1, "flat"))                 # the reference's flat pass == ours, byte for byte
    rays = O.viewport(72, 54, [("mov", (0.1, 0.05, -0.2)), ("rot", (0.0, 0.15, 0.0))])
    assert np.array_equal(load(1, "pt"), O.render_counter(rays, t, m, 3, 1)[0])
    assert np.array_equal(load(0, "pt"), O.render_mt(rays, 72, 54, t, m, 3, 8))
