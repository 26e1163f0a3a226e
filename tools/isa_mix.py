#!/usr/bin/env python
"""Instruction mix of the MFMA loops of one kernel in a hipcc --save-temps .s file.  usage: isa_mix.py file.s <symbol-substring>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
print(lines[start][:120])
blocks, cur, lab = [], [], "entry"
for l in body:
    if re.match(r"\.LBB\d+_\d+:", l):
        blocks.append((lab, cur)); cur, lab = [], l
    else:
        cur.append(l)
blocks.append((lab, cur))
for lab, b in blocks:
    ins = [x.strip().split()[0] for x in b if x.startswith("\t") and not x.strip().startswith((".", ";"))]
    if sum("mfma" in i for i in ins) < 4:
        continue
    c = Counter()
    for i in ins:
        if "mfma" in i: c["mfma"] += 1
        elif i.startswith("ds_"): c[i] += 1
        elif i.startswith(("global_", "buffer_", "scratch_")): c[i] += 1
        elif i in ("s_waitcnt", "s_barrier", "s_nop"): c[i] += 1
        elif i.startswith("s_"): c["salu"] += 1
        elif i.startswith("v_"): c["valu"] += 1
        else: c[i] += 1
    print("  ", lab, len(ins), dict(c))
for l in lines[end:end + 60]:
    m = re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", l)
    if m: print("  ", m.group(1), m.group(2))
