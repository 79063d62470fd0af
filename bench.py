#!/usr/bin/env python3
"""Headline benchmark: Mray/s (primary x spp x bounces) of the path-tracing hot path on MI355X.

A step = one full render (renderer::render semantics, reference src/cpu_renderer.cpp:29-79) of
BASELINE.json configs[2]: closed-room scene with 10,000 triangles, 1920x1080, 256 spp, 5 bounces,
all inputs (rays, triangles, materials) resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
N > 1 without a torch.distributed environment: this process starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py`, before anything touches
the GPU) and relays rank 0's JSON line; started under torch.distributed.run it is a rank.

N > 1: the same image is split into pixel-row tiles dealt round-robin to the ranks (strong
scaling; spath_amd/dist.py); each step ends with one RCCL gather of the RGBA8 tiles to rank 0.

Prints ONE JSON line on rank 0 with the driver's contract fields plus `roofline` and
`cpu_baseline`:
  roofline            the most loaded pipe of the dominant kernel: FP32 vector (VALU) instruction issue.
                      achieved = executed lane-instructions per second = ray-triangle tests/s x lane-instructions
                      per test, the latter from a committed rocprofv3 PMC pass of this very command
                      (profiles/valu_issue.json <- tools/valu_issue_from_pmc.py <- SQ_INSTS_VALU);
                      peak = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (MI355X_MICROARCH.md).  frac <= 1.
  roofline.mfma       the default scan (rpl_cylm) evaluates stage 1's side products on the f16 matrix pipe: executed matrix
                      flop/s (32 per test) against the 2.5 PFLOP/s dense f16 peak -- the less loaded of the two pipes.
  roofline.hbm_effective   the accounting SURVEY.md 8(d) / the north_star name: scans x n_tris x 48 B
                      (sizeof(geom::triangle), what the reference's scan reads per ray) / kernel time against the
                      8 TB/s HBM peak.  It is an EFFECTIVE bandwidth and exceeds 1: every triangle fetched into LDS
                      serves the 256 rays of a workgroup, the 640 KB record stream lives in L2; `traffic` is what HBM really moved.
Kernel time = HIP events on the launch stream around every step's kernels.
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
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 matrix peak (~2.5 PF)
VALU_MEASURED_FMAC_TINSTR = 56.7   # measured on MI355X, 4 waves/SIMD, independent v_fmac_f32 chains (profiles/r01_valu_microbench_extended.log)
VALU_PEAK_TINSTR = 256 * 4 * 32 * 2.4e9 / 1e12   # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.64 T lane-instructions/s
# Only f32 add / mul / fma issue at that rate.  Measured at the kernel's 4 waves per SIMD (tools/valu_bench2.hip, profiles/r03_valu_microbench_classes.log):
# v_fma_f32 55.2, v_sub_f32 57.7, v_fmac_f32 53.7 T lane-instr/s -- and 34.4 - 35.8 for v_minimum3_f32, v_min3_f32, v_min_f32, v_max_f32, v_alignbit_b32,
# v_lshl_or_b32: 16 lanes/clk, 4 cycles per wave64 instruction.  The mix of the kernel comes from the per-type SQ counters (valu_issue.json "classes").
VALU_CLASS_CYCLES = {"full": 2.0, "half": 4.0, "trans": 8.0}             # nominal issue cycles per wave64 instruction
VALU_CLASS_MEASURED_TINSTR = {"full": 55.2, "half": 34.5, "trans": 18.2}  # sustained at 4 waves/SIMD (trans: v_rcp_f32, profiles/r01_valu_microbench_extended.log)
BYTES_PER_TEST = 48          # sizeof(geom::triangle)
FLOPS_PER_TEST = 52


def self_launch(argv, n):
    """`python bench.py --gpus N` with no torch.distributed environment: start the N ranks as children of this
    process (which never touches the GPU), pass rank 0's JSON line through, fail if any rank fails."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    other = [l for l in p.stdout.splitlines() if l.strip() and not l.startswith("{")]
    if other:
        print("\n".join(other), file=sys.stderr)
    if p.returncode != 0 or not lines:
        print(f"bench.py: the {n}-rank launch failed (rc {p.returncode})", file=sys.stderr)
        return p.returncode or 1
    print(lines[-1], flush=True)
    return 0


class ClockSampler:
    """Shader clock while the timed steps run, from the driver's pp_dpm_sclk table (the '*' row), every 0.25 s.
    The guide's caveat applies: sysfs may read a few % above the in-kernel clock; this is context for the roofline."""

    def __init__(self, pci_bus_id=None):
        import glob
        self.paths = []
        if pci_bus_id:                    # the device this rank renders on, e.g. 0000:75:00.0
            self.paths = glob.glob(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/pp_dpm_sclk")
        self.samples, self._stop, self._t = [], False, None

    def _read(self, path):
        try:
            for line in open(path).read().splitlines():
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        except Exception:
            return None
        return None

    def start(self):
        import threading
        if not self.paths:
            return
        path = self.paths[0]

        def loop():
            while not self._stop:
                v = self._read(path)
                if v:
                    self.samples.append(v)
                time.sleep(0.25)
        self._t = threading.Thread(target=loop, daemon=True)
        self._t.start()

    def stop(self):
        self._stop = True
        if self._t:
            self._t.join(timeout=1.0)
        return (sum(self.samples) / len(self.samples)) if self.samples else None


def profile_entry(fname, key, lib_hash=None):
    """Figures that only a profiler pass can supply (PMC counters, the statistics build), committed under profiles/.
    Every entry is stamped with the source/flags hash of the library it was measured on (sphip_build_info); an entry whose stamp
    differs from the loaded library's is NOT used: returns (None, reason)."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        e = json.load(open(path)).get(key)
    except Exception as ex:
        return None, f"profiles/{fname} unreadable ({ex})"
    if e is None:
        return None, f"no entry {key!r} in profiles/{fname}: no profiler pass committed for this configuration and kernel"
    if lib_hash is not None and e.get("source_hash") != lib_hash:
        return None, (f"profiles/{fname}[{key!r}] was measured on sources {e.get('source_hash', 'unstamped')}, the loaded library is built from "
                      f"{lib_hash}: stale, not used (re-run tools/collect_profiles.sh + tools/publish_profiles.sh)")
    return e, None


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
    quantum = 0.0
    if O.have_ref():
        try:
            _, info = O.ref_run("render", w, h, spp, tris, mats, threads=cores, return_info=True)
        except Exception as e:                 # e.g. the prebuilt binary does not run on this host: time the port instead
            print(f"bench.py: oracle/_ref/spath_ref failed ({e}); timing the C restatement", file=sys.stderr)
            info = None
    if info and "seconds" in info:
        secs, kind = info["seconds"], "reference"
        what = "unmodified reference cpu_renderer (oracle/_ref/spath_ref)"
        # render_pt_mt returns when its 250 ms completion poll next fires (reference src/cpu_renderer.cpp:172-178): the measured
        # time is the compute time rounded UP to the poll grid -- what a caller of the reference waits, and up to 0.25 s more than the work
        quantum = 0.25
    else:
        rays = view.Camera(w, h).get_viewport()
        O.lib()
        t0 = time.perf_counter()
        O.render_mt(rays, w, h, tris, mats, spp, cores, cores)
        secs, kind = time.perf_counter() - t0, "port"
        what = "C restatement of cpu_renderer (oracle/liboracle.so)"
    return {"value": nominal / secs / 1e6, "unit": "Mray/s", "cores": cores, "kind": kind,
            "seconds": round(secs, 3), "seconds_poll_quantum": quantum,
            "value_range": [round(nominal / secs / 1e6, 5), round(nominal / max(secs - quantum, 1e-9) / 1e6, 5)],
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
    ap.add_argument("--no-worst-case", action="store_true", help="skip the (untimed) large-triangle scene figure")
    ap.add_argument("--no-valu-microbench", action="store_true", help="do not run build/valu_bench2 (the per-class VALU issue rates of THIS box, ~1 s, untimed)")
    ap.add_argument("--cpu-w", type=int, default=192)
    ap.add_argument("--cpu-h", type=int, default=108)
    ap.add_argument("--cpu-spp", type=int, default=32, help="the CPU sample's spp (default: ~30 s of the reference on 16 threads, 0.25 s poll quantum < 1 %)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from spath_amd import capi, scene, view
    from spath_amd.dist import Gatherer, RowTilePlan, ShardedRenderer, balanced_tile_rows

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
    gather = Gatherer(plan, rank)
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    image = None
    gather_s = []                      # host time of the exchange of every timed step (N > 1), this rank

    def step(i_timed=None):
        nonlocal image
        if i_timed is not None:
            ev[i_timed][0].record()
        local = shard.render(SPP, seed=1, mode=capi.MODE_PT, flags=flags, stream=stream)
        if i_timed is not None:
            ev[i_timed][1].record()
        if world > 1:
            if i_timed is not None:
                torch.cuda.synchronize()           # (the gather needs the finished tiles anyway; this separates kernel from exchange time)
                tg = time.perf_counter()
            image = gather(local.cpu() if rehearsal else local)
            if i_timed is not None:
                torch.cuda.synchronize()
                gather_s.append(time.perf_counter() - tg)
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
    bus = None
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        bus = None
    clock = ClockSampler(bus)
    if rank == 0:
        clock.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    sclk_mhz = clock.stop() if rank == 0 else None
    # per-channel sums of the assembled RGBA8 frame (identical for every --gpus N: pixel-keyed RNG); taken now, the untimed
    # runs below reuse the shard's output buffer
    image_sum = [int(x) for x in image[:, :3].to(torch.int64).sum(dim=0).tolist()] if rank == 0 else None
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
        # (two-stage kernels: a closest-hit pre-pass, one scan per pixel, then the same path-tracing kernel from bounce 1 on)
        spp_r = SPP
        e0.record()
        shard.render(spp_r, seed=1, mode=capi.MODE_PT, flags=flags | capi.FLAG_PRIMARY_REUSE, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        sr = ctx.stats()
        exact_only["with_primary_reuse"] = {"spp": spp_r, "kernel_ms": round(e0.elapsed_time(e1), 3),
                                            "nominal_Mray_per_s": round(W * H * spp_r * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e6, 2),
                                            "scans_executed": sr["scans_executed"], "nominal_scans": W * H * spp_r * 5}
        # and the OPT-IN acceleration structure (linear BVH): a different work definition, never the headline
        spp_r = min(SPP, 16)
        e0.record()
        shard.render(spp_r, seed=1, mode=capi.MODE_PT, flags=capi.FLAG_ACCEL, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        exact_only["with_accel_structure_opt_in"] = {"spp": spp_r, "kernel_ms": round(e0.elapsed_time(e1), 3),
                                                     "nominal_Mray_per_s": round(W * H * spp_r * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e6, 2)}
    # the VALU issue rates of this box by instruction class (untimed, a child process, GPU otherwise idle): tools/valu_bench2.hip, built by
    # __graft_entry__.build() into build/valu_bench2; without it the committed rates of profiles/r03_valu_microbench_classes.log are used
    live_rates = None
    vb2 = os.path.join(ROOT, "build", "valu_bench2")
    if world == 1 and not args.no_valu_microbench and os.path.exists(vb2):
        try:
            import subprocess
            out = subprocess.run([vb2], capture_output=True, text=True, timeout=120).stdout
            rate = {}
            for line in out.splitlines():
                if "waves/SIMD=4" in line:
                    rate[line.split("waves/SIMD=4")[0].strip()] = float(line.split("waves/SIMD=4")[1].split()[0]) / 1e3    # T lane-instr/s
            full = [v for k, v in rate.items() if k.startswith(("v_fma_f32", "v_sub_f32", "v_fmac_f32"))]
            half = [v for k, v in rate.items() if k.startswith(("v_minimum3_f32", "v_alignbit_b32", "v_min_f32", "v_max_f32", "v_min3_f32", "v_lshl_or_b32"))]
            if full and half:
                live_rates = {"full": round(sum(full) / len(full), 2), "half": round(sum(half) / len(half), 2),
                              "stage1_block": round(rate.get("stage-1 block: 2 x (minimum3 x2, fma, sub, alignbit)", 0.0), 2),
                              "source": "build/valu_bench2 (tools/valu_bench2.hip) run by this bench on this GPU, 4 waves per SIMD"}
        except Exception as e:           # a measurement aid, never a reason to fail the bench
            live_rates = {"error": repr(e)}
    worst = None
    if world == 1 and NT >= 64 and not args.no_worst_case:
        # large triangles (clutter x10: most rays cross most cylinders): how far the two-stage scan degrades.  Untimed, after
        # the timed region, with the same default kernel (a rocprofv3 pass of this command: --no-worst-case keeps the profile unmixed)
        wt, wm = scene.closed_room(NT, clutter_scale=10.0)
        d_wt, d_wm = torch.from_numpy(wt).to(dev), torch.from_numpy(wm).to(dev)
        ctx.set_scene_device(d_wt.data_ptr(), d_wm.data_ptr(), NT, stream)
        spp_w = min(SPP, 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        shard.render(spp_w, seed=1, mode=capi.MODE_PT, flags=flags, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        sw = ctx.stats()
        worst = {"scene": f"closed_room({NT}, clutter_scale=10): clutter triangles 10x larger",
                 "kernel": capi.load().sphip_kernel_name(sw["kernel_variant"]).decode(), "spp": spp_w,
                 "kernel_ms": round(e0.elapsed_time(e1), 3), "Mray_per_s": round(W * H * spp_w * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e6, 2),
                 "tests_per_s": round(sw["scans_executed"] * NT / (e0.elapsed_time(e1) * 1e-3), 1),
                 "note": "no survivor queue, hence no overflow path: degrades smoothly towards the exact-only scan"}
        ctx.set_scene_device(d_tris.data_ptr(), d_mats.data_ptr(), NT, stream)
        torch.cuda.synchronize()
    if world > 1:
        cdev = torch.device("cpu") if rehearsal else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([float(st["scans_executed"]), max(kernel_ms) if kernel_ms else 0.0, 1e3 * sum(gather_s) / max(len(gather_s), 1)],
                         dtype=torch.float64, device=cdev)
        s_sum = s.clone(); dist.all_reduce(s_sum, op=dist.ReduceOp.SUM)
        s_max = s.clone(); dist.all_reduce(s_max, op=dist.ReduceOp.MAX)
        s_min = s.clone(); dist.all_reduce(s_min, op=dist.ReduceOp.MIN)
        scans_per_step, kern_ms_max, kern_ms_min = float(s_sum[0].item()), float(s_max[1].item()), float(s_min[1].item())
        gather_ms_max, gather_ms_min = float(s_max[2].item()), float(s_min[2].item())
    else:
        scans_per_step = float(st["scans_executed"])
        kern_ms_max = kern_ms_min = (sum(kernel_ms) / len(kernel_ms) if kernel_ms else 0.0)
        gather_ms_max = gather_ms_min = 0.0

    if rank == 0:
        nominal = W * H * SPP * 5
        ms_per_step = elapsed * 1e3 / max(args.steps, 1)
        value = nominal * args.steps / elapsed / 1e6
        # dominant kernel = the path-trace kernel (one launch per step per rank); average launch duration
        avg_kernel_s = (sum(kernel_ms) / len(kernel_ms) * 1e-3) if kernel_ms else float("nan")
        my_scans = float(st["scans_executed"])
        achieved = my_scans * NT * BYTES_PER_TEST / avg_kernel_s / 1e9
        kname = capi.load().sphip_kernel_name(st["kernel_variant"]).decode()
        key = f"{NT}tris_{W}x{H}x{SPP}_g{world}_{kname}"
        lib_hash = capi.build_source_hash()
        tr, tr_why = profile_entry("hbm_traffic.json", key, lib_hash)
        if tr is None and world > 1:
            tr, tr_why = profile_entry("hbm_traffic.json", f"{NT}tris_{W}x{H}x{SPP}_g1_{kname}", lib_hash)
            tr = None                              # (a 1-GPU launch's bytes are not a shard's bytes: reported as unavailable)
        traffic = (tr or {}).get("hbm_bytes_per_launch")
        vi, vi_why = profile_entry("valu_issue.json", key, lib_hash)
        if vi is None and world > 1:
            # instructions per ray-triangle test are a property of the kernel and the scene, not of the GPU count: rank 0's shard of
            # an N-GPU run executes the same kernel on the same scene as the committed 1-GPU PMC pass
            vi, vi_why = profile_entry("valu_issue.json", f"{NT}tris_{W}x{H}x{SPP}_g1_{kname}", lib_hash)
            if vi:
                vi = dict(vi)
                vi["source"] = vi.get("source", "") + " [1-GPU pass of the same kernel and scene, applied to rank 0's shard]"
        vi = vi or {}
        fs, _ = profile_entry("filter_stats.json", f"{NT}tris_{W}x{H}_{kname}", lib_hash)
        fs = fs or {}
        tests_per_s = my_scans * NT / avg_kernel_s
        lane_instr = vi.get("lane_instr_per_test")
        valu_achieved = tests_per_s * lane_instr / 1e12 if lane_instr else None
        # the same instructions priced by issue class (see VALU_CLASS_*): what share of the SIMDs' issue time the kernel's own mix needs
        issue_classes = None
        cls = vi.get("classes") if vi else None
        if cls and valu_achieved and cls.get("source_hash") == lib_hash:
            mix = {"full": cls["full_rate_frac"], "half": cls["half_rate_frac"], "trans": cls["trans_frac"]}
            cyc = sum(mix[k] * VALU_CLASS_CYCLES[k] for k in mix)                         # nominal issue cycles per wave64 instruction of this mix
            rates = dict(VALU_CLASS_MEASURED_TINSTR)
            if live_rates and "full" in live_rates:
                rates["full"], rates["half"] = live_rates["full"], live_rates["half"]       # this box, this run
            sec_per_tinstr = sum(mix[k] / rates[k] for k in mix)     # seconds per 1e12 lane-instructions at the measured class rates
            issue_classes = {
                "full_rate_frac": mix["full"], "half_rate_frac": mix["half"], "trans_frac": mix["trans"],
                "nominal_cycles_per_instr": round(cyc, 3),
                "frac_class_weighted": round(valu_achieved / (VALU_PEAK_TINSTR * 2.0 / cyc), 4),
                "frac_of_measured_class_rates": round(valu_achieved * sec_per_tinstr, 4),
                "class_rates_tinstr": {k: rates[k] for k in ("full", "half", "trans")},
                "class_rates_source": (live_rates or {}).get("source", "profiles/r03_valu_microbench_classes.log (committed; build/valu_bench2 not present or not run)"),
                "note": "only f32 add/mul/fma issue at the 32 lanes/clk of `peak`; min/max, integer, compare/select, bit and cross-lane instructions issue at 16 "
                        "(measured: profiles/r03_valu_microbench_classes.log).  frac_class_weighted prices every instruction at its nominal issue cycles (2 / 4 / 8); "
                        "frac_of_measured_class_rates at the rates a pure stream of each class sustains at this kernel's 4 waves per SIMD",
                "source": cls.get("source"),
            }
        elif cls:
            issue_classes = {"note": "class mix measured on another build of the library: not applied", "measured_on": cls.get("source_hash")}
        out = {
            "metric": "Mray/s (primary x spp x bounces)",
            "value": round(value, 3),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (stage-1 filter: f16 hi/lo products on the matrix pipe, f32 accumulate)",
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
            "image_sum_rgb": image_sum,
            "nominal_rays_per_step": nominal,
            "reference_runs_untimed": exact_only,
            "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
            "kernel_ms_per_rank": {"min": round(kern_ms_min, 3), "max": round(kern_ms_max, 3)},
            "gather_ms_per_step": ({"min": round(gather_ms_min, 3), "max": round(gather_ms_max, 3),
                                    "note": "host time of the RCCL gather of the finished tiles + un-permute on rank 0, per timed step, slowest / fastest rank"}
                                   if world > 1 else None),
            "library_source_hash": lib_hash,
            "worst_case_untimed": worst,
            "roofline": {
                # the binding roof: FP32 vector instruction issue (HBM is idle, see `traffic`; the f16 matrix pipe carries stage 1's side
                # products in rpl_cylm but is the less loaded of the two pipes, see `mfma`)
                "bound": "valu",
                "achieved": round(valu_achieved, 3) if valu_achieved else None,
                "peak": round(VALU_PEAK_TINSTR, 3),
                "unit": "Tinstr/s",
                "frac": round(valu_achieved / VALU_PEAK_TINSTR, 4) if valu_achieved else None,
                "traffic": traffic,
                "kernel": "k_pt_filter<R, SPLIT, SCAN> (" + kname + ")" if st["kernel_variant"] >= 3 else "k_pt<variant>",
                "avg_kernel_ms": round(avg_kernel_s * 1e3, 3),
                "tests_per_s": round(tests_per_s, 1),
                "lane_instr_per_test": lane_instr,
                "lane_instr_source": vi.get("source", vi_why),
                "traffic_source": (tr or {}).get("method", tr_why) if traffic is None else tr.get("method"),
                "peak_definition": "256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (MI355X_MICROARCH.md); one wave64 VALU instruction = 2 issue cycles",
                # what a pure stream of independent v_fmac_f32 issues on this chip at the kernel's 4 waves per SIMD (tools/valu_bench.hip,
                # profiles/r01_valu_microbench_extended.log: 56.7 T lane-instr/s; v_fma_f32 50-55, three-source min/med ops 31.5)
                "issue_classes": issue_classes,
                "measured_fmac_issue_rate": VALU_MEASURED_FMAC_TINSTR,
                "frac_of_measured_fmac_rate": round(valu_achieved / VALU_MEASURED_FMAC_TINSTR, 4) if valu_achieved else None,
                "sclk_observed_mhz": round(sclk_mhz, 1) if sclk_mhz else None,
                "frac_at_observed_sclk": round(valu_achieved / (256 * 4 * 32 * sclk_mhz * 1e6 / 1e12), 4) if (valu_achieved and sclk_mhz) else None,
                "filter": {"pairs_surviving_stage1": fs.get("survivor_frac"), "stage2_rounds_per_tile": fs.get("rounds_per_tile"),
                           "stage2_lane_utilisation": fs.get("lane_utilisation"), "queue_overflows": 0 if st["kernel_variant"] >= 9 else fs.get("overflows"),
                           "source": fs.get("source")},
                # the north_star's accounting: logical triangle-stream bandwidth against the HBM peak (exceeds 1 by design)
                "hbm_effective": {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                                  "algorithmic_bytes_per_launch": my_scans * NT * BYTES_PER_TEST,
                                  "note": "effective (logical-stream) bandwidth of rank 0's launch; exceeds the HBM peak because a record fetched "
                                          "into LDS serves every ray of the workgroup and the stream lives in L2; HBM really moved `traffic` bytes"},
                "algorithmic_flop_frac": round(my_scans * NT * FLOPS_PER_TEST / avg_kernel_s / (FP32_PEAK_TFLOPS * 1e12), 4),
                # rpl_cylm: one v_mfma_f32_32x32x16_f16 (32768 flop) per 32 x 32 ray-triangle tests = 32 executed f16 flop per test
                "mfma": ({"achieved": round(tests_per_s * 32.0 / 1e12, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(tests_per_s * 32.0 / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                          "note": "executed flop of the matrix instruction (K = 16: 15 half products + 1 for P_a per test), dense f16 peak"}
                         if kname == "rpl_cylm" else None),
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
