#!/bin/bash
# usage (GPU box, repo root): tools/quick_check.sh <tag> [pytest -k expr]   -> gradient parity tests + one bench line with the per-kernel table
tag=$1; kexpr=${2:-"grads or fullsize_oracle or e2e"}
mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -15 > gpurun_out/$tag/pytest.log; tail -4 gpurun_out/$tag/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-env > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo bench rc $?
python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
r=d["roofline"]
print("dominant", r["kernel"], r["frac"], "min_frac", r["min_frac"], "whole", r["whole_step"]["frac"])
for k in r["kernels"]: print("  %-12s %7.1f us %6.1f TF  frac %.3f  share %.3f" % (k["kernel"], k["avg_us"], k["achieved"], k["frac"], k["time_share"]))
PY
