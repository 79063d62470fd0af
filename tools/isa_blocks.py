#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing: python tools/isa_blocks.py build/cur.s 'k_hit_filter<4, 1>' [min_instrs]"""
import re, subprocess, sys
from collections import Counter
txt = open(sys.argv[1]).read()
want = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for n in re.findall(r"^(_Z\S+):", txt, re.M):
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    if want in d:
        body = txt[txt.index(n + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        blocks, cur, name = [], [], "entry"
        for l in body.split("\n"):
            if re.match(r"^\.LBB\d+_\d+:", l):
                blocks.append((name, cur)); name = l.split(":")[0]; cur = []
            else:
                cur.append(l)
        blocks.append((name, cur))
        for nm, b in blocks:
            ins = [l.strip() for l in b if l.strip() and not l.strip().startswith((";", "."))]
            if len(ins) < mn: continue
            c = Counter(i.split()[0] for i in ins)
            g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
            print(f"{nm:12s} n {len(ins):4d} valu {g('v_'):4d} fma {sum(v for k,v in c.items() if 'fma' in k):3d} cndmask {g('v_cndmask'):3d} ds_r {g('ds_read'):2d} ds_w {g('ds_write'):2d} glob {g('global_'):2d} scratch {g('scratch_'):2d} waitcnt {g('s_waitcnt'):2d} branch {g('s_cbranch'):2d}")
        break
