#!/bin/bash
# usage (GPU box, repo root): tools/bench_prof.sh <tag> [grep pattern] [extra bench args]  -> gpurun_out/<tag>/kernel_stats.md  (isolated per-kernel times: the profiler serialises the queues)
tag=$1; pat=${2:-.}; shift; shift
mkdir -p gpurun_out/$tag; R=$PWD
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/prof -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/$tag/run.log 2>&1
cd $R
python tools/rocprof_summary.py $(find gpurun_out/$tag/prof -name "*.db" | head -1) > gpurun_out/$tag/kernel_stats.md 2>&1
grep -E "$pat" gpurun_out/$tag/kernel_stats.md | cut -c1-160 | head -${HEAD:-40}
