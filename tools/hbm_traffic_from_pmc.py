"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a one-step bench.py run into profiles/hbm_traffic.json.
usage: python tools/hbm_traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <key> [round [source hash of the measured library]]
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled on gfx950 as MI355X_MICROARCH.md prescribes."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as _ge
SOURCE_HASH = _ge.source_hash()      # the sources/flags the measured library was built from (run this right after the measurement)


def total(path, counter):
    per = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            name = row["Kernel_Name"].split("(")[0]
            per[name] = per.get(name, 0.0) + float(row["Counter_Value"])
    return per


fetch, write = total(sys.argv[1], "FETCH_SIZE"), total(sys.argv[2], "WRITE_SIZE")
dom = max((k for k in fetch if "k_pt" in k), key=lambda k: fetch[k] + write.get(k, 0.0))
res = [k for k in fetch if "k_resolve" in k]
entry = {
    "hbm_bytes_per_launch": int(fetch[dom] * 2 * 1024 + write.get(dom, 0.0) * 1024),
    "fetch_size_kb_raw": fetch[dom], "write_size_kb_raw": write.get(dom, 0.0), "kernel": dom.replace("void ", ""),
    "resolve_pass_hbm_bytes": int(sum(fetch[k] * 2 * 1024 + write.get(k, 0.0) * 1024 for k in res)) if res else 0,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` "
              "(profiles/r02_bench_pmc_*.csv); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; the reads "
              "here are 8-B and 4-B per lane, for which the guide calls the factor uncalibrated, so this is an upper bound). The traffic is the "
              "path-history work buffer (8 B per scan written and read back) and the per-sample radiance scratch of the sample-chunked launch, "
              "not triangle data: the 320 KB + 480 KB record streams stay in L2/LDS",
    "round": int(sys.argv[4]) if len(sys.argv) > 4 else 3, "source_hash": sys.argv[5] if len(sys.argv) > 5 else SOURCE_HASH,
}
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
allj = json.load(open(path)) if os.path.exists(path) else {}
allj[sys.argv[3]] = entry
json.dump(allj, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
