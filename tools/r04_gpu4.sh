set -x
mkdir -p gpurun_out/r4d
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4d/rc.txt
timeout 300 python tools/actor_probe.py 20 > gpurun_out/r4d/actor.log 2>&1
ALGO=impala timeout 300 python tools/actor_probe.py 20 >> gpurun_out/r4d/actor.log 2>&1
timeout 300 python tools/pipeline_probe.py > gpurun_out/r4d/probe.log 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4d/prof -o a -- python $R/tools/actor_probe.py 5 > $R/gpurun_out/r4d/prof.log 2>&1
cd $R; python tools/rocprof_summary.py $(find gpurun_out/r4d/prof -name "*.db" | head -1) > gpurun_out/r4d/actor_kernel_stats.md 2>&1
rm -rf gpurun_out/r4d/prof
cat gpurun_out/r4d/rc.txt; tail -8 gpurun_out/r4d/pytest.log; cat gpurun_out/r4d/actor.log gpurun_out/r4d/probe.log; head -12 gpurun_out/r4d/actor_kernel_stats.md | cut -c1-150
