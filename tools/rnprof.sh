# usage (GPU box): tools/rnprof.sh [iters]  -> per-kernel table of tools/rn_microbench.py (one rollout + isolated ResNet learner minibatches)
R=$PWD; mkdir -p gpurun_out/rnprof; export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/rnprof/tr -o t -- python $R/tools/rn_microbench.py ${1:-4} 2>&1 | grep minibatch; cd $R
python tools/rocprof_summary.py $(find gpurun_out/rnprof/tr -name "*.db" | head -1) > gpurun_out/rnprof/kernel_stats.md; rm -rf gpurun_out/rnprof/tr
awk -F'|' 'NR>2 && $3+0 < 60 {printf "%-76s %4s %8s\n", $2, $3, $5}' gpurun_out/rnprof/kernel_stats.md | head -45
