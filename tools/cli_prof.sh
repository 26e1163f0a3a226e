#!/bin/bash
# usage (GPU box, repo root): tools/cli_prof.sh <tag> <module: cleanba_ppo|cleanba_impala> [cli args]  -> gpurun_out/<tag>/kernel_stats.md
tag=$1; mod=$2; shift; shift
mkdir -p gpurun_out/$tag; R=$PWD
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/prof -o cli -- python -m cleanba_amd.$mod --env-backend device --local-num-envs 120 --num-actor-threads 1 --log-frequency 1 "$@" > $R/gpurun_out/$tag/run.log 2>&1
cd $R
grep -v "^W2026\|^E2026" gpurun_out/$tag/run.log | tail -3
python tools/rocprof_summary.py $(find gpurun_out/$tag/prof -name "*.db" | head -1) > gpurun_out/$tag/kernel_stats.md 2>&1
head -${HEAD:-24} gpurun_out/$tag/kernel_stats.md | cut -c1-150; tail -1 gpurun_out/$tag/kernel_stats.md
