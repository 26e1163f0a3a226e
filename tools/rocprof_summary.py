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


def exclusive(con):
    """Per kernel name: average of (end - max(start, end of the previous dispatch on the same queue)).  A dispatch's own duration runs from
    its first wave to its last, so a kernel that starts while its predecessor on the SAME in-order queue is still draining (the dispatcher
    overlaps back-to-back launches; one-block-per-CU kernels have long drains) is charged the overlap twice; the exclusive time is what the
    bench's HIP events measure (an event pair serialises the two launches)."""
    cols = [d[0] for d in con.execute("select * from kernels limit 1").description]
    q = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
    if q is None or "start" not in cols or "end" not in cols:
        return {}
    last, acc = {}, {}
    for name, st, en, qu in con.execute(f"select name, start, end, {q} from kernels order by start"):
        ex = en - max(st, last.get(qu, 0))
        last[qu] = max(en, last.get(qu, 0))
        a = acc.setdefault(name, [0, 0])
        a[0] += max(ex, 0); a[1] += 1
    return {n: a[0] / a[1] for n, a in acc.items()}


def main(path):
    con = sqlite3.connect(path)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name").fetchall()
    tot = sum(r[2] for r in rows)
    try:
        ex = exclusive(con)
    except sqlite3.Error:
        ex = {}
    print(f"| kernel | calls | total ms | avg us | min us | max us | % | avg us after the predecessor on its queue ended |\n|---|---|---|---|---|---|---|---|")
    for n, c, s, a, mn, mx in sorted(rows, key=lambda r: -r[2]):
        e = f"{ex[n] / 1e3:.1f}" if n in ex else ""
        print(f"| {short(n)} | {c} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / tot:.1f} | {e} |")
    print(f"\ntotal kernel time {tot / 1e6:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
