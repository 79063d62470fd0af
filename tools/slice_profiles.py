"""Counters of a declared configs[3] / configs[4] slice (tools/slice_once.py under rocprofv3 --pmc) -> profiles/valu_issue.json and
profiles/hbm_traffic.json, keyed by the slice's own key.
usage: python tools/slice_profiles.py <slice json line file> <sq counter_collection.csv> <fetch csv> <write csv> [tcc csv] [round]
The profiled process launches the path-tracing kernel twice (1 spp to build the streams, then the slice): the LAST dispatch is used."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
info = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rnd = int(sys.argv[6]) if len(sys.argv) > 6 else 3


def last_dispatch(path):
    """{counter: value} of the last k_pt* dispatch in a counter_collection.csv"""
    rows = [r for r in csv.DictReader(open(path)) if "k_pt" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    out, name = {}, None
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    return out, name


sq, kern = last_dispatch(sys.argv[2])
fetch, _ = last_dispatch(sys.argv[3])
write, _ = last_dispatch(sys.argv[4])
tcc = last_dispatch(sys.argv[5])[0] if len(sys.argv) > 5 and os.path.exists(sys.argv[5]) else {}
tests = info["tests"]
vi = {"lane_instr_per_test": round(sq["SQ_INSTS_VALU"] * 64 / tests, 4), "kernel": kern, "launches": 1, "tests": tests,
      "counters": {k: v for k, v in sorted(sq.items())}, "round": rnd, "source_hash": info["library_source_hash"],
      "kernel_ms_under_pmc": info["kernel_ms"],
      "source": f"rocprofv3 --pmc pass of `python3 tools/slice_once.py {info['slice']} {info['spp']}` (profiles/r{rnd:02d}_{info['slice']}_pmc_sq_counters.csv), last dispatch: "
                "SQ_INSTS_VALU x 64 / (scans x n_tris)"}
for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD"):
    if k in sq:
        vi[k.lower() + "_per_test_x64"] = round(sq[k] * 64 / tests, 4)
stream_bytes = info["n_tris"] * (64 + 48)
tr = {"hbm_bytes_per_launch": int(fetch["FETCH_SIZE"] * 2 * 1024 + write["WRITE_SIZE"] * 1024), "fetch_size_kb_raw": fetch["FETCH_SIZE"], "write_size_kb_raw": write["WRITE_SIZE"],
      "kernel": kern, "round": rnd, "source_hash": info["library_source_hash"],
      "record_streams_bytes": stream_bytes,
      "stream_reads_if_every_workgroup_scan_missed": int(info["scans_executed"] / 256 * info["n_tris"] * 64),
      "method": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python3 tools/slice_once.py {info['slice']} {info['spp']}`, last dispatch; FETCH_SIZE doubled "
                "per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; exact for the 16-B-per-lane LDS-DMA of the tile stream, an upper bound for the "
                "narrower reads of the work buffer)"}
if tcc:
    tr["tcc_hit_sum"], tr["tcc_miss_sum"] = tcc.get("TCC_HIT_sum"), tcc.get("TCC_MISS_sum")
    if tcc.get("TCC_HIT_sum") is not None and tcc.get("TCC_MISS_sum") is not None:
        tr["l2_hit_rate"] = round(tcc["TCC_HIT_sum"] / max(tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"], 1.0), 4)
for fname, entry in (("valu_issue.json", vi), ("hbm_traffic.json", tr)):
    path = os.path.join(ROOT, "profiles", fname)
    allj = json.load(open(path)) if os.path.exists(path) else {}
    allj[info["key"]] = entry
    json.dump(allj, open(path, "w"), indent=1)
print(info["key"]); print(json.dumps(vi, indent=1)); print(json.dumps(tr, indent=1))
