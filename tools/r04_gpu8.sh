set -x
mkdir -p gpurun_out/r4g
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4g/rc.txt
( time timeout 900 python bench.py > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err ) 2> gpurun_out/r4g/bench.time
cat gpurun_out/r4g/rc.txt; tail -4 gpurun_out/r4g/pytest.log; cat gpurun_out/r4g/bench.time
