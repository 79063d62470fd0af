"""Issue-class mix of the dominant kernel's VALU instructions, from a rocprofv3 --pmc pass of bench.py with the per-type counters.
usage: python tools/valu_classes_from_pmc.py <counter_collection.csv> <bench line of that run (.json)> [round]
  full rate  = SQ_INSTS_VALU_ADD_F32 + MUL_F32 + FMA_F32   (the SIMD issues these at 32 lanes/clk: 2 cycles per wave64 instruction)
  trans      = SQ_INSTS_VALU_TRANS_F32                      (8 cycles)
  half rate  = everything else (min/max, integer, compare/select, bit and cross-lane operations; 16 lanes/clk: 4 cycles) --
               which instructions sit in which class is measured by tools/valu_bench2.hip (profiles/r03_valu_microbench_classes.log)
Adds valu_issue.json[key]["classes"], key as tools/valu_issue_from_pmc.py; bench.py turns it into the class-weighted issue fractions."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(sys.argv[1])))
bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
rnd = int(sys.argv[3]) if len(sys.argv) > 3 else 3
per = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_pt" in name:
        per.setdefault(name, {}).setdefault(r["Counter_Name"], 0.0)
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
dom = max(per, key=lambda k: per[k].get("SQ_INSTS_VALU", 0.0))
c = per[dom]
total = c["SQ_INSTS_VALU"]
full = c.get("SQ_INSTS_VALU_ADD_F32", 0.0) + c.get("SQ_INSTS_VALU_MUL_F32", 0.0) + c.get("SQ_INSTS_VALU_FMA_F32", 0.0)
trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
entry = {
    "full_rate_frac": round(full / total, 4), "trans_frac": round(trans / total, 4), "half_rate_frac": round(1.0 - (full + trans) / total, 4),
    "int32_frac": round(c.get("SQ_INSTS_VALU_INT32", 0.0) / total, 4), "cvt_frac": round(c.get("SQ_INSTS_VALU_CVT", 0.0) / total, 4),
    "kernel": dom, "counters": {k: v for k, v in sorted(c.items())},
    "source": f"rocprofv3 --pmc pass of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` with the per-type VALU counters "
              f"(profiles/{os.path.basename(sys.argv[1])}); class rates: tools/valu_bench2.hip (profiles/r03_valu_microbench_classes.log)",
    "round": rnd, "source_hash": bench.get("library_source_hash"),
}
cfg = bench["config"]
key = f"{cfg['n_tris']}tris_{cfg['width']}x{cfg['height']}x{cfg['spp']}_g{bench['n_gpus']}_{cfg['kernel']}"
path = os.path.join(ROOT, "profiles", "valu_issue.json")
allj = json.load(open(path)) if os.path.exists(path) else {}
if key not in allj:
    sys.exit(f"no entry {key} in profiles/valu_issue.json: run tools/valu_issue_from_pmc.py first")
if allj[key].get("source_hash") != entry["source_hash"]:
    sys.exit(f"{key}: the instruction-count entry was measured on another build ({allj[key].get('source_hash')} != {entry['source_hash']})")
allj[key]["classes"] = entry
json.dump(allj, open(path, "w"), indent=1)
print(key, json.dumps(entry, indent=1))
