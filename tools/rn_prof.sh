#!/bin/bash
# usage (on the GPU box, from repo root): tools/rn_prof.sh <tag> [extra cli args]  -> gpurun_out/<tag>/kernel_stats.md
tag=$1; shift
mkdir -p gpurun_out/$tag; R=$PWD
timeout 300 python -m pytest tests/test_gpu_resnet.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/prof -o rn -- python -m cleanba_amd.cleanba_ppo --network impala_resnet --env-backend device --local-num-envs 120 --num-steps 128 --num-actor-threads 1 --total-timesteps 61440 --log-frequency 1 "$@" > $R/gpurun_out/$tag/run.log 2>&1
cd $R
grep -v "^W2026\|^E2026" gpurun_out/$tag/run.log | tail -3
python tools/rocprof_summary.py $(find gpurun_out/$tag/prof -name "*.db" | head -1) > gpurun_out/$tag/kernel_stats.md 2>&1
head -${HEAD:-30} gpurun_out/$tag/kernel_stats.md | cut -c1-150; tail -1 gpurun_out/$tag/kernel_stats.md
