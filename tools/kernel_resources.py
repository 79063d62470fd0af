#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel in a hipcc -S listing (default build/cur.s)."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "build/cur.s"
pat = sys.argv[2] if len(sys.argv) > 2 else ""
txt = open(path).read()
names = re.findall(r"^\s*\.amdhsa_kernel (\S+)", txt, re.M)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
for n, d in zip(names, dem):
    blk = txt[txt.index(".amdhsa_kernel " + n):]
    blk = blk[:blk.index(".end_amdhsa_kernel")]
    tail = txt[txt.index(".end_amdhsa_kernel", txt.index(".amdhsa_kernel " + n)):][:3000]
    g = lambda k: (re.search(r"; %s: (\d+)" % k, tail) or [None, "?"])[1]
    if pat in d:
        print(f"{d[:100]:100s} vgpr {g('NumVgprs'):>3} sgpr {g('NumSgprs'):>3} scratch {g('ScratchSize'):>4} lds {g('LDSByteSize'):>6} occ {g('Occupancy')}")
