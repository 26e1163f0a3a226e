#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace results.db into a per-kernel stats table (markdown)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"void (igemm\w*)_kernel<(\w+)<IgemmTile<(\d+), (\d+), (\d+), (\d+), (\d+)(?:, \d+)?>(.*)", name)
    if m:
        extra = re.sub(r"[<> ]", "", m.group(8))[:24]
        return f"{m.group(1)}<{m.group(2)} {m.group(3)}x{m.group(4)}x{m.group(5)} {extra}>"
    return name[:70]


def main(path):
    con = sqlite3.connect(path)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for n, c, s, a, mn, mx in sorted(rows, key=lambda r: -r[2]):
        print(f"| {short(n)} | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / tot:.1f} |")
    print(f"\ntotal kernel time {tot / 1e6:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
