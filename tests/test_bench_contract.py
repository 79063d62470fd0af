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
    assert d["unit"] == "Mray/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"].startswith("f32")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # the binding roof is FP32 vector issue: peak 256 CU x 4 SIMD x 32 lanes x 2.4 GHz; the fraction needs a committed PMC pass for
    # the configuration (absent for this toy workload: null, never a number above 1)
    assert r["bound"] == "valu" and r["unit"] == "Tinstr/s" and abs(r["peak"] - 78.643) < 1e-2
    assert r["frac"] is None or (0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3)
    h = r["hbm_effective"]
    assert h["unit"] == "GB/s" and h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-3
    assert r["filter"]["queue_overflows"] == 0 and d["rccl_ranks"] == 1
    assert d["worst_case_untimed"]["Mray_per_s"] > 0
    assert abs(d["value"] - d["nominal_rays_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3 + 1e-4 / d["ms_per_step"]    # (rounding of the printed ms)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mray/s" and "sample" in c
    assert d["scans_per_step"] <= d["nominal_rays_per_step"]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_matches_one_rank():
    """The N > 1 code path (row-tile sharding, gather, max-over-ranks timing) rehearsed on one GPU with gloo."""
    common = ["--steps", "1", "--warmup", "0", "--width", "160", "--height", "90", "--spp", "3", "--tris", "300", "--no-cpu-baseline"]
    one = _run(common)
    # `python bench.py --gpus 2` with no torch.distributed environment: bench.py starts its own two ranks
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SPATH_BENCH_REHEARSAL"] = "1"
    two = _run(["--gpus", "2"] + common, env=env)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["rccl_ranks"] == 2
    assert two["image_sum_rgb"] == one["image_sum_rgb"] and two["scans_per_step"] == one["scans_per_step"]
    assert two["kernel_ms_per_rank"]["min"] <= two["kernel_ms_per_rank"]["max"]


def test_bench_self_launch_reports_a_failing_rank():
    """No GPU here: the two ranks bench.py starts exit non-zero, and so must the parent (no hang, no JSON line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, cwd=ROOT)
    assert p.returncode != 0 and "needs a GPU" in p.stderr
