"""bench.py prints one JSON line with the driver's contract fields plus roofline and cpu_baseline (tiny workload)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _run(args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_small_workload():
    d = _run(["--steps", "2", "--warmup", "1", "--width", "256", "--height", "144", "--spp", "4", "--tris", "500",
              "--cpu-w", "32", "--cpu-h", "18", "--cpu-spp", "2"])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mray/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - d["nominal_rays_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mray/s" and "sample" in c
    assert d["scans_per_step"] <= d["nominal_rays_per_step"]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_matches_one_rank():
    """The N > 1 code path (row-tile sharding, gather, max-over-ranks timing) rehearsed on one GPU with gloo."""
    common = ["--steps", "1", "--warmup", "0", "--width", "160", "--height", "90", "--spp", "3", "--tris", "300", "--no-cpu-baseline"]
    one = _run(common)
    env = dict(os.environ, SPATH_BENCH_REHEARSAL="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                       capture_output=True, text=True, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    two = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["image_sum_rgb"] == one["image_sum_rgb"] and two["scans_per_step"] == one["scans_per_step"]


def test_bench_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, cwd=ROOT)
    assert p.returncode != 0 and "needs a GPU" in p.stderr
