#!/usr/bin/env python
"""How fast does the rollout advance under each learner kernel?  Reads a `rocprofv3 --kernel-trace` results.db of tools/pipeline_probe.py (or any
pipelined PPO run) and prints, per learner kernel: calls, average duration, total time and the number of ACTOR kernels that completed per
millisecond while it ran.  Alone the actor completes ~85 kernels per ms (641 per 128-step rollout in 7.5 ms); the step time of the pipelined
trainer follows from sum(time_in_kernel * rate) = 641.
usage (GPU box):  cd /tmp; rocprofv3 --kernel-trace -d out -o t -- python $REPO/tools/pipeline_probe.py;  python tools/actor_progress.py $(find out -name '*.db')"""
import bisect
import re
import sqlite3
import sys


def is_actor(n):
    return any(k in n for k in ("igemm_s16", "actor_tail", "env_step", "sample_kernel"))


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void ", "")
    m = re.match(r"(igemm\w*)_kernel<(\w+)<IgemmTile<(\d+), (\d+), (\d+)", n)
    return f"{m.group(1)}<{m.group(2)} {m.group(3)}x{m.group(4)}x{m.group(5)}>" if m else n[:48]


def main(path):
    con = sqlite3.connect(path)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    act = [(s, e) for n, s, e in rows if is_actor(n)]
    lrn = [(s, e, short(n)) for n, s, e in rows if not is_actor(n)]
    g = [s for s, e, n in lrn if n.startswith("gae_kernel")]
    # the pipelined phase of pipeline_probe.py = updates 9..13 (the first eight run after their rollout finished); any other trace: everything
    lo, hi = (g[8], g[13]) if len(g) >= 14 else (rows[0][1], rows[-1][2])
    lrn = [x for x in lrn if lo <= x[0] < hi]
    act = [x for x in act if lo <= x[0] < hi]
    ends = sorted(e for s, e in act)
    stat = {}
    for s, e, n in lrn:
        k = bisect.bisect_right(ends, e) - bisect.bisect_right(ends, s)
        a = stat.setdefault(n, [0, 0.0, 0])
        a[0] += 1; a[1] += e - s; a[2] += k
    tot_t = sum(a[1] for a in stat.values()); tot_k = sum(a[2] for a in stat.values())
    print(f"learner kernel time {tot_t / 1e6:.1f} ms; actor kernels completed inside learner kernels: {tot_k} of {len(act)}\n")
    print("| learner kernel | calls | avg us | total ms | actor kernels completed per ms |\n|---|---|---|---|---|")
    for n, a in sorted(stat.items(), key=lambda x: -x[1][1])[:20]:
        print("| %s | %d | %.1f | %.2f | %.1f |" % (n, a[0], a[1] / a[0] / 1e3, a[1] / 1e6, a[2] / (a[1] / 1e6)))


if __name__ == "__main__":
    main(sys.argv[1])
