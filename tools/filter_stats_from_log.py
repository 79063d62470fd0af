"""Survivor statistics of the two-stage scan from a -DSP_FILTER_STATS build's stderr line -> profiles/filter_stats.json.
usage: python tools/filter_stats_from_log.py <log> <n_tris> <W> <H> <kernel name> <source note>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as _ge
txt = open(sys.argv[1]).read()
nt, w, h, kname = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
m = re.search(r"\[cyl stats\] group bits set=(\d+) stage-2 rounds\(per wave\)=(\d+) wave-tiles=(\d+) exact tests=(\d+)", txt)
scans = int(re.search(r"scans (\d+)", txt).group(1))
bits, rounds, tiles, exact = (int(x) for x in m.groups())
entry = {"survivor_frac": round(exact / (scans * nt), 6), "group_bits_per_lane_and_tile": round(bits / (64 * tiles), 3),
         "rounds_per_tile": round(rounds / tiles, 3), "lane_utilisation": round(exact / (64 * rounds), 4), "overflows": 0,
         "exact_tests": exact, "pairs": scans * nt, "source": sys.argv[6],
         # (the statistics build is the same sources with -DSP_FILTER_STATS: stamped with the hash of the shipped build of those sources)
         "source_hash": _ge.source_hash()}
m2 = re.search(r"wave-wide exact turns=(\d+) \(([0-9.]+) per round, ([0-9.]+) of their lanes used\)", txt)
if m2:
    entry["exact_turns_per_tile"] = round(int(m2.group(1)) / tiles, 3)
    entry["exact_turn_lane_utilisation"] = float(m2.group(3))
path = os.path.join(ROOT, "profiles", "filter_stats.json")
allj = json.load(open(path)) if os.path.exists(path) else {}
allj[f"{nt}tris_{w}x{h}_{kname}"] = entry
json.dump(allj, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
