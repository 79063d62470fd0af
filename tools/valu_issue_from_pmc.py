"""VALU lane-instructions per ray-triangle test of the dominant kernel, from a rocprofv3 --pmc pass of bench.py.
usage: python tools/valu_issue_from_pmc.py <counter_collection.csv> <bench line of that run (.json)> [round]
  lane_instr_per_test = SQ_INSTS_VALU (wave instructions issued, summed over the dominant kernel's launches) x 64
                        / (launches x scans_per_step x n_tris)
Writes profiles/valu_issue.json[key], key = '<tris>tris_<W>x<H>x<spp>_g<gpus>_<kernel>', which bench.py reads."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as _ge
SOURCE_HASH = _ge.source_hash()      # the sources/flags the measured library was built from (run this right after the measurement)
rows = list(csv.DictReader(open(sys.argv[1])))
bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
rnd = int(sys.argv[3]) if len(sys.argv) > 3 else 3
SOURCE_HASH = bench.get("library_source_hash") or SOURCE_HASH     # the library the PMC pass itself ran on, when the bench line says
per, launches = {}, {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_pt" in name:
        per.setdefault(name, {}).setdefault(r["Counter_Name"], 0.0)
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        launches.setdefault(name, set()).add(r["Dispatch_Id"])
dom = max(per, key=lambda k: per[k].get("SQ_INSTS_VALU", 0.0))
c = bench["config"]
n = len(launches[dom])
tests = n * bench["scans_per_step"] * c["n_tris"]
entry = {
    "lane_instr_per_test": round(per[dom]["SQ_INSTS_VALU"] * 64 / tests, 4),
    "kernel": dom, "launches": n, "tests": tests,
    "counters": {k: v for k, v in sorted(per[dom].items())},
    "source": f"rocprofv3 --pmc pass of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` (profiles/{os.path.basename(sys.argv[1])}): "
              f"SQ_INSTS_VALU x 64 / (scans x n_tris); reproduce with tools/pmc_summary.py",
    "round": rnd, "source_hash": SOURCE_HASH,
}
for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD"):
    if k in per[dom]:
        entry[k.lower() + "_per_test_x64"] = round(per[dom][k] * 64 / tests, 4)
key = f"{c['n_tris']}tris_{c['width']}x{c['height']}x{c['spp']}_g{bench['n_gpus']}_{c['kernel']}"
path = os.path.join(ROOT, "profiles", "valu_issue.json")
allj = json.load(open(path)) if os.path.exists(path) else {}
allj[key] = entry
json.dump(allj, open(path, "w"), indent=1)
print(key, json.dumps(entry, indent=1))
