R=$PWD; mkdir -p gpurun_out/span; export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/span/tr -o t -- python $R/tools/pipeline_probe.py 2>&1 | grep "ms"; cd $R
python - <<'PY'
import sqlite3, glob, bisect, re
db = glob.glob("gpurun_out/span/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
def is_actor(n): return any(k in n for k in ("igemm_s16", "actor_tail", "env_step", "sample_kernel"))
def short(n):
    n = re.sub(r"\(.*$", "", n); n = n.replace("void ", "")
    m = re.match(r"(igemm\w*)_kernel<(\w+)<IgemmTile<(\d+), (\d+), (\d+)", n)
    return f"{m.group(1)}<{m.group(2)} {m.group(3)}x{m.group(4)}x{m.group(5)}>" if m else n[:48]
act = [(s, e) for n, s, e in rows if is_actor(n)]
lrn = [(s, e, short(n)) for n, s, e in rows if not is_actor(n)]
# pipelined phase only: where actor kernels overlap learner ones: take learner kernels between 9th and 14th gae
g = [s for s, e, n in lrn if n.startswith("gae_kernel")]
lo, hi = g[8], g[13]
lrn = [x for x in lrn if lo <= x[0] < hi]
act = [x for x in act if lo <= x[0] < hi]
ends = sorted(e for s, e in act)
stat = {}
for s, e, n in lrn:
    k = bisect.bisect_right(ends, e) - bisect.bisect_right(ends, s)   # actor kernels that COMPLETED while this learner kernel ran
    a = stat.setdefault(n, [0, 0.0, 0])
    a[0] += 1; a[1] += e - s; a[2] += k
tot_t = sum(a[1] for a in stat.values()); tot_k = sum(a[2] for a in stat.values())
print(f"learner kernel time {tot_t/1e6:.1f} ms, actor kernels completed inside learner kernels {tot_k} of {len(act)}")
print("%-52s %6s %9s %9s %12s" % ("learner kernel", "calls", "avg us", "total ms", "actor k / ms"))
for n, a in sorted(stat.items(), key=lambda x: -x[1][1])[:24]:
    print("%-52s %6d %9.1f %9.2f %12.1f" % (n, a[0], a[1] / a[0] / 1e3, a[1] / 1e6, a[2] / (a[1] / 1e6)))
PY
rm -rf gpurun_out/span/tr
