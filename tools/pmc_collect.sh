#!/bin/bash
# usage (GPU box, repo root): tools/pmc_collect.sh <tag>
#   1. rocprofv3 --kernel-trace --stats over `python bench.py` — the headline workload only: the host_env / secondary / cpu_baseline legs
#      launch the same kernels at other sizes and would mix into the per-kernel averages      -> gpurun_out/<tag>/kernel_stats.md
#   2. four SEPARATE rocprofv3 --pmc passes over one isolated learner minibatch (tools/microbench.py --plain), as MI355X_MICROARCH.md
#      prescribes (never --pmc together with the sys/hip/hsa trace domains)                          -> gpurun_out/<tag>/pmc_summary.md, pmc_traffic.json
tag=$1; R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-env --no-secondary --no-baseline-config > $out/trace.log 2>&1
python $R/tools/rocprof_summary.py $(find $out/trace -name "*.db" | head -1) > $out/kernel_stats.md 2>&1
pass() { timeout 400 rocprofv3 --pmc $2 --kernel-trace -d $out/pmc_$1 -o p -- python $R/tools/microbench.py 3 --plain > $out/pmc_$1.log 2>&1; }
pass sq "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
pass lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
db() { find $out/pmc_$1 -name "*.db" | head -1; }
cd $R
python tools/microbench.py 1 --plain --names-out $out/kernel_names.json > /dev/null 2>&1
python tools/pmc_report.py $(db sq) $(db lds) $(db fetch) $(db write) > $out/pmc_summary.md 2>&1
python tools/pmc_traffic.py $(db fetch) $(db write) $out/kernel_names.json > $out/pmc_traffic.json 2>$out/pmc_traffic.err
rm -rf $out/trace $out/pmc_sq $out/pmc_lds $out/pmc_fetch $out/pmc_write   # the .db files are tens of MB; the summaries are what gets committed
head -30 $out/kernel_stats.md | cut -c1-150; cat $out/pmc_summary.md | cut -c1-170 | tail -22; head -c 600 $out/pmc_traffic.json
