#!/usr/bin/env python
"""Per-kernel PMC averages from rocprofv3 --pmc results.db files (counters summed over XCD instances per dispatch)."""
import re
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import short


def main(paths, filt="igemm"):
    table = defaultdict(dict)
    for path in paths:
        con = sqlite3.connect(path)
        q = ("select name, counter_name, dispatch_id, sum(counter_value), max(duration) from pmc_events group by name, counter_name, dispatch_id")
        acc = defaultdict(list)
        dur = defaultdict(list)
        for name, cn, did, val, d in con.execute(q):
            if filt and filt not in name:
                continue
            acc[(short(name), cn)].append(val)
            dur[short(name)].append(d)
        for (n, cn), vals in acc.items():
            big = [v for v in vals]
            table[n][cn] = sum(big) / len(big)
        for n, ds in dur.items():
            table[n].setdefault("dur_us", sum(ds) / len(ds) / 1e3)
    cols = sorted({c for v in table.values() for c in v})
    print("| kernel | " + " | ".join(cols) + " |")
    print("|---|" + "---|" * len(cols))
    for n, v in sorted(table.items(), key=lambda kv: -kv[1].get("dur_us", 0)):
        print(f"| {n} | " + " | ".join(f"{v.get(c, float('nan')):.4g}" for c in cols) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
