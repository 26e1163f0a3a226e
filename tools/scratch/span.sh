R=$PWD; mkdir -p gpurun_out/span; export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/span/tr -o t -- python $R/tools/pipeline_probe.py 2>&1 | grep "ms"; cd $R
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/span/tr/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
cols = [d[0] for d in con.execute("select * from kernels limit 1").description]
rows = con.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
def is_actor(n): return any(k in n for k in ("igemm_s16", "actor_tail", "env_step", "sample_kernel"))
act = [(s, e) for n, s, e in rows if is_actor(n)]
lrn = [(s, e, n) for n, s, e in rows if not is_actor(n)]
# rollouts: split actor kernels at gaps > 0.5 ms
spans = []; cur = [act[0][0], act[0][1], 0.0, 0]
for s, e in act[1:]:
    if s - cur[1] > 500e3: spans.append(cur); cur = [s, e, 0.0, 0]
    cur[1] = max(cur[1], e); cur[2] += e - s; cur[3] += 1
spans.append(cur)
print("rollout spans (start ms, span ms, kernel-time ms, kernels):")
for sp in spans[:14]: print(f"  {(sp[0]-t0)/1e6:9.2f} {(sp[1]-sp[0])/1e6:8.2f} {sp[2]/1e6:8.2f} {sp[3]}")
# learner updates: split at gae_kernel
g = [s for s, e, n in lrn if "gae_kernel" in n]
ad = [e for s, e, n in lrn if "adam" in n]
print("gae starts (ms):", [round((x - t0) / 1e6, 2) for x in g[:14]])
import itertools
# last adam end before next gae
ends = []
for i, gs in enumerate(g):
    nxt = g[i + 1] if i + 1 < len(g) else 1e30
    es = [e for e in ad if gs < e < nxt]
    if es: ends.append((gs, max(es)))
print("update spans ms:", [round((e - s) / 1e6, 2) for s, e in ends[:14]])
PY
rm -rf gpurun_out/span/tr
