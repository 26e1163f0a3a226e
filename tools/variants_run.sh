#!/bin/bash
# GPU box: bench every variant built by tools/variants.sh (plus the shipped library as "base") and print the per-kernel table side by side.
mkdir -p gpurun_out/abl; export TMPDIR=/tmp
for name in base "$@"; do
  so=cleanba_amd/abl_$name.so; [ $name = base ] && so=cleanba_amd/libcleanba_mi.so
  CBM_SO=$PWD/$so timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-env > gpurun_out/abl/$name.json 2> gpurun_out/abl/$name.err || echo "$name FAILED"
done
python - "$@" <<'PY'
import json, sys
names = ["base"] + sys.argv[1:]
rows = {}
for n in names:
    try:
        d = json.loads(open(f"gpurun_out/abl/{n}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "no result", e); continue
    rows[n] = d
print("%-12s" % "kernel" + "".join("%12s" % n[:11] for n in rows))
print("%-12s" % "ms/step" + "".join("%12.3f" % rows[n]["ms_per_step"] for n in rows))
ks = [k["kernel"] for k in next(iter(rows.values()))["roofline"]["kernels"]]
for i, k in enumerate(ks):
    print("%-12s" % k + "".join("%12.1f" % rows[n]["roofline"]["kernels"][i]["avg_us"] for n in rows))
PY
