#!/usr/bin/env python3
"""Headline benchmark: Mray/s (primary x spp x bounces) of the path-tracing hot path on MI355X.

A step = one full render (renderer::render semantics, reference src/cpu_renderer.cpp:29-79) of
BASELINE.json configs[2]: closed-room scene with 10,000 triangles, 1920x1080, 256 spp, 5 bounces,
all inputs (rays, triangles, materials) resident in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: the same image is split into pixel-row tiles dealt round-robin to the ranks (strong
scaling; spath_amd/dist.py); each step ends with one RCCL gather of the RGBA8 tiles to rank 0.

Prints ONE JSON line on rank 0 with the driver's contract fields plus `roofline` and
`cpu_baseline`:
  roofline.achieved = scans_executed x n_tris x 48 B / kernel time  (algorithmic bytes the
  reference's scan reads per ray, sizeof(geom::triangle), geom.h:185-190), kernel time from HIP
  events on the launch stream.  It is an *effective* bandwidth: LDS/L2 reuse lets it exceed the
  HBM peak, which is why valu_frac (52 flop per ray-triangle test against the 157.3 TFLOP/s
  FP32 vector peak) is printed next to it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector
BYTES_PER_TEST = 48          # sizeof(geom::triangle)
FLOPS_PER_TEST = 52


def host_cores() -> int:
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    env = os.environ.get("SPATH_CPU_THREADS")
    return int(env) if env else n


def cpu_baseline(tris, mats, args):
    """The reference's cpu_renderer timed on this box's host cores on a bounded sample of the same scene."""
    import numpy as np
    from oracle import oracle as O
    from spath_amd import view
    cores = host_cores()
    w, h, spp = args.cpu_w, args.cpu_h, args.cpu_spp
    nominal = w * h * spp * 5
    info = None
    if O.have_ref():
        try:
            _, info = O.ref_run("render", w, h, spp, tris, mats, threads=cores, return_info=True)
        except Exception as e:                 # e.g. the prebuilt binary does not run on this host: time the port instead
            print(f"bench.py: oracle/_ref/spath_ref failed ({e}); timing the C restatement", file=sys.stderr)
            info = None
    if info and "seconds" in info:
        secs, kind = info["seconds"], "reference"
        what = "unmodified reference cpu_renderer (oracle/_ref/spath_ref)"
    else:
        rays = view.Camera(w, h).get_viewport()
        O.lib()
        t0 = time.perf_counter()
        O.render_mt(rays, w, h, tris, mats, spp, cores, cores)
        secs, kind = time.perf_counter() - t0, "port"
        what = "C restatement of cpu_renderer (oracle/liboracle.so)"
    return {"value": nominal / secs / 1e6, "unit": "Mray/s", "cores": cores, "kind": kind,
            "seconds": round(secs, 3),
            "sample": f"{what}, same scene ({tris.shape[0]} triangles), {w}x{h} x {spp} spp x 5 bounces, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--tris", type=int, default=10000)
    ap.add_argument("--tile-rows", type=int, default=0, help="rows per tile of the round-robin shard plan; 0 = largest height <= 8 that balances the ranks")
    ap.add_argument("--kernel", type=str, default="auto", help="scan kernel variant (see sphip_kernel_name)")
    ap.add_argument("--primary-reuse", action="store_true", help="scan the primary ray once per pixel (fewer scans; off for roofline runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true", help="after the timed region also run (untimed, reduced spp) the exact-only scan, "
                    "primary-hit reuse and the opt-in acceleration structure, and report them under reference_runs_untimed; "
                    "off by default so that a profile of the default command contains only the timed kernel")
    ap.add_argument("--cpu-w", type=int, default=192)
    ap.add_argument("--cpu-h", type=int, default=108)
    ap.add_argument("--cpu-spp", type=int, default=16)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from spath_amd import capi, scene, view
    from spath_amd.dist import RowTilePlan, ShardedRenderer, balanced_tile_rows, gather_to_root

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # rehearsal on a 1-GPU box: SPATH_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two
    # ranks on one device); the real multi-GPU run is one rank per GPU over RCCL ("nccl" backend on ROCm)
    rehearsal = os.environ.get("SPATH_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    W, H, SPP, NT = args.width, args.height, args.spp, args.tris
    tris, mats = scene.closed_room(NT)
    rays = view.Camera(W, H).get_viewport()
    if args.tile_rows <= 0:
        args.tile_rows = balanced_tile_rows(H, world)
    plan = RowTilePlan(W, H, world, args.tile_rows)

    ctx = capi.Context(dev_index)
    variants = capi.kernel_variants()
    if args.kernel not in variants:
        raise SystemExit(f"unknown --kernel {args.kernel}; have {sorted(variants)}")
    flags = variants[args.kernel] | (capi.FLAG_PRIMARY_REUSE if args.primary_reuse else 0)
    stream = torch.cuda.current_stream().cuda_stream
    d_tris, d_mats = torch.from_numpy(tris).to(dev), torch.from_numpy(mats).to(dev)
    ctx.set_scene_device(d_tris.data_ptr(), d_mats.data_ptr(), NT, stream)
    shard = ShardedRenderer(ctx, plan, rank, rays, dev)
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    image = None

    def step(i_timed=None):
        nonlocal image
        if i_timed is not None:
            ev[i_timed][0].record()
        local = shard.render(SPP, seed=1, mode=capi.MODE_PT, flags=flags, stream=stream)
        if i_timed is not None:
            ev[i_timed][1].record()
        if world > 1:
            image = gather_to_root(local.cpu() if rehearsal else local, plan, rank)
        else:
            image = local

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()                       # figures of the last launch on this rank
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    # outside the timed region, for reference: the exact-only scan (every pair through the full Moeller-Trumbore test,
    # no conservative pre-test) on the same frame at 8 spp -- same image bits, ~4x the instructions per test
    exact_only = None
    if world == 1 and NT >= 64 and args.extras:
        spp_x = min(SPP, 8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        shard.render(spp_x, seed=1, mode=capi.MODE_PT, flags=variants["rpl_lds"], stream=stream)
        e1.record()
        torch.cuda.synchronize()
        sx = ctx.stats()
        exact_only = {"kernel": "rpl_lds", "spp": spp_x, "kernel_ms": round(e0.elapsed_time(e1), 3),
                      "Mray_per_s": round(W * H * spp_x * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e6, 2),
                      "tests_per_s": round(sx["scans_executed"] * NT / (e0.elapsed_time(e1) * 1e-3), 1)}
        # and the default kernel with primary-hit reuse (the primary ray of a pixel is scanned once instead
        # of once per sample -- identical image, ~20 % fewer scans); NOT used for the headline figure
        spp_r = min(SPP, 16)
        e0.record()
        shard.render(spp_r, seed=1, mode=capi.MODE_PT, flags=flags | capi.FLAG_PRIMARY_REUSE, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        sr = ctx.stats()
        exact_only["with_primary_reuse"] = {"spp": spp_r, "kernel_ms": round(e0.elapsed_time(e1), 3),
                                            "nominal_Mray_per_s": round(W * H * spp_r * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e6, 2),
                                            "scans_executed": sr["scans_executed"], "nominal_scans": W * H * spp_r * 5}
        # and the OPT-IN acceleration structure (linear BVH): a different work definition, never the headline
        e0.record()
        shard.render(spp_r, seed=1, mode=capi.MODE_PT, flags=capi.FLAG_ACCEL, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        exact_only["with_accel_structure_opt_in"] = {"spp": spp_r, "kernel_ms": round(e0.elapsed_time(e1), 3),
                                                     "nominal_Mray_per_s": round(W * H * spp_r * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e6, 2)}
    if world > 1:
        cdev = torch.device("cpu") if rehearsal else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([float(st["scans_executed"]), max(kernel_ms) if kernel_ms else 0.0], dtype=torch.float64, device=cdev)
        s_sum = s.clone(); dist.all_reduce(s_sum, op=dist.ReduceOp.SUM)
        s_max = s.clone(); dist.all_reduce(s_max, op=dist.ReduceOp.MAX)
        scans_per_step, kern_ms_max = float(s_sum[0].item()), float(s_max[1].item())
    else:
        scans_per_step, kern_ms_max = float(st["scans_executed"]), (sum(kernel_ms) / len(kernel_ms) if kernel_ms else 0.0)

    if rank == 0:
        nominal = W * H * SPP * 5
        ms_per_step = elapsed * 1e3 / max(args.steps, 1)
        value = nominal * args.steps / elapsed / 1e6
        # dominant kernel = the path-trace kernel (one launch per step per rank); average launch duration
        avg_kernel_s = (sum(kernel_ms) / len(kernel_ms) * 1e-3) if kernel_ms else float("nan")
        my_scans = float(st["scans_executed"])
        achieved = my_scans * NT * BYTES_PER_TEST / avg_kernel_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{NT}tris_{W}x{H}x{SPP}_g{world}"
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mray/s (primary x spp x bounces)",
            "value": round(value, 3),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"closed-room {NT} triangles, {W}x{H}, {SPP} spp, 5 bounces (BASELINE.json configs[2])"
                            if (NT, W, H, SPP) == (10000, 1920, 1080, 256) else f"closed-room {NT} triangles, {W}x{H}, {SPP} spp, 5 bounces",
                "n_tris": NT, "width": W, "height": H, "spp": SPP, "bounces": 5,
                "sharding": "whole image on 1 GPU" if world == 1 else f"{args.tile_rows}-row tiles round-robin over {world} GPUs + RCCL gather",
                "kernel": capi.load().sphip_kernel_name(st["kernel_variant"]).decode(),
                "primary_reuse": bool(args.primary_reuse),
                "arithmetic": "strict (no FMA contraction, IEEE divide): bit-identical to the CPU oracle",
            },
            "scans_per_step": scans_per_step,
            # per-channel sums of the assembled RGBA8 frame: identical for every --gpus N (pixel-keyed RNG)
            "image_sum_rgb": [int(x) for x in image[:, :3].to(torch.int64).sum(dim=0).tolist()],
            "nominal_rays_per_step": nominal,
            "reference_runs_untimed": exact_only,
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": "k_pt_filter<R>" if st["kernel_variant"] >= 3 else "k_pt<variant>",
                "avg_kernel_ms": round(avg_kernel_s * 1e3, 3),
                "algorithmic_bytes_per_launch": my_scans * NT * BYTES_PER_TEST,
                "note": "effective (logical-stream) bandwidth of rank 0's launch; exceeds HBM peak because triangles are reused from LDS/L2",
                "tests_per_s": round(my_scans * NT / avg_kernel_s, 1),
                "valu_frac": round(my_scans * NT * FLOPS_PER_TEST / avg_kernel_s / (FP32_PEAK_TFLOPS * 1e12), 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tris, mats, args)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
