set -x
mkdir -p gpurun_out/r4c
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c/rc.txt
timeout 300 python tools/actor_probe.py 20 > gpurun_out/r4c/actor.log 2>&1
ALGO=impala timeout 300 python tools/actor_probe.py 20 >> gpurun_out/r4c/actor.log 2>&1
timeout 300 python tools/pipeline_probe.py > gpurun_out/r4c/probe.log 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4c/prof -o a -- python $R/tools/actor_probe.py 5 > $R/gpurun_out/r4c/prof.log 2>&1
cd $R; python tools/rocprof_summary.py $(find gpurun_out/r4c/prof -name "*.db" | head -1) > gpurun_out/r4c/actor_kernel_stats.md 2>&1
rm -rf gpurun_out/r4c/prof
cat gpurun_out/r4c/rc.txt; tail -5 gpurun_out/r4c/pytest.log; cat gpurun_out/r4c/actor.log gpurun_out/r4c/probe.log; head -12 gpurun_out/r4c/actor_kernel_stats.md | cut -c1-150
