#!/usr/bin/env python
"""profiles/rNN_pmc_summary.md from four rocprofv3 --pmc passes over tools/microbench.py (see the header it prints).
usage: pmc_report.py <sq.db> <lds.db> <fetch.db> <write.db>"""
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import short


def load(path):
    con = sqlite3.connect(path)
    acc, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for name, cn, did, val, d in con.execute("select name, counter_name, dispatch_id, sum(counter_value), max(duration) from pmc_events group by name, counter_name, dispatch_id"):
        acc[name][cn].append(val)
        dur[name].append(d)
    return {n: {c: sum(v) / len(v) for c, v in cs.items()} for n, cs in acc.items()}, {n: sum(v) / len(v) / 1e3 for n, v in dur.items()}


sq, dur = load(sys.argv[1])
lds, _ = load(sys.argv[2])
fe, _ = load(sys.argv[3])
wr, _ = load(sys.argv[4])
print("rocprofv3 --pmc <set> --kernel-trace -- python tools/microbench.py 3 --plain   (MI355X; four separate passes: {SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES "
      "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU}, {SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS "
      "SQ_INSTS_VMEM_RD}, {FETCH_SIZE}, {WRITE_SIZE}; learner kernels of one 3840-frame PPO minibatch, isolated)\n")
print("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); wait share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; LDS conflict share = SQ_LDS_BANK_CONFLICT / "
      "SQ_LDS_IDX_ACTIVE; fetch = 2 x FETCH_SIZE KiB (gfx950 correction, L2 misses incl. Infinity-Cache hits).\n")
print("| kernel | us | MfmaUtil | wave wait share | LDS conflict share | VALU insts per MFMA-busy cycle x64 | fetch MB | write MB |")
print("|---|---|---|---|---|---|---|---|")
for n, d in sorted(dur.items(), key=lambda kv: -kv[1]):
    if d < 8 or n not in sq:
        continue
    s, l = sq[n], lds.get(n, {})
    mf = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (d * 1e-6 * 2.4e9 * 1024)
    wait = s.get("SQ_WAIT_INST_ANY", 0) / max(s.get("SQ_WAVE_CYCLES", 1), 1)
    conf = l.get("SQ_LDS_BANK_CONFLICT", 0) / max(l.get("SQ_LDS_IDX_ACTIVE", 1), 1)
    vpm = s.get("SQ_INSTS_VALU", 0) / max(s.get("SQ_VALU_MFMA_BUSY_CYCLES", 1), 1) * 64
    vtxt = f"{vpm:.2f}" if mf > 0.01 else "-"
    print(f"| {short(n)[:64]} | {d:.1f} | {100 * mf:.1f} % | {100 * wait:.1f} % | {100 * conf:.1f} % | {vtxt} | {2 * fe.get(n, {}).get('FETCH_SIZE', 0) / 1024:.0f} | "
          f"{wr.get(n, {}).get('WRITE_SIZE', 0) / 1024:.0f} |")
