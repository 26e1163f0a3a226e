#!/bin/bash
# Reproduction loop for the multi-process IPC shapes on ONE GPU: runs `bench.py --gpus N` (dp line, then the BASELINE topology line) K times with
# the HIP / KFD error logs on, one stdout / stderr pair per iteration; prints one verdict line per iteration and a total.
# usage: tools/flaky_loop.sh <outdir> <iterations> <gpus> [extra env assignments...]
out=${1:-gpurun_out/flaky}; iters=${2:-20}; gpus=${3:-8}; shift 3
mkdir -p "$out"
ok=0
for i in $(seq 1 "$iters"); do
  t0=$(date +%s)
  env CBM_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 AMD_LOG_LEVEL=${AMD_LOG_LEVEL:-1} HSAKMT_DEBUG_LEVEL=${HSAKMT_DEBUG_LEVEL:-3} "$@" \
    timeout 600 python bench.py --gpus "$gpus" --steps 2 --warmup 1 --no-cpu-baseline --no-host-env > "$out/out_$i.json" 2> "$out/err_$i.log"
  rc=$?
  v=$(python - "$out/out_$i.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    bc = d.get("baseline_config") or {}
    print("dp_value=%s dp_error=%s topo_value=%s topo_error=%s" % (d.get("value"), d.get("error"), bc.get("value"), (bc.get("error") or "")[:300]))
    sys.exit(0 if d.get("value") and bc.get("value") else 1)
except Exception as e:
    print("unparsable:", e); sys.exit(1)
PY
)
  good=$?
  [ $good -eq 0 ] && [ $rc -eq 0 ] && ok=$((ok+1))
  echo "iter $i rc=$rc $(( $(date +%s) - t0 ))s $v" | tee -a "$out/summary.txt"
done
echo "TOTAL $ok/$iters green (gpus=$gpus)" | tee -a "$out/summary.txt"
