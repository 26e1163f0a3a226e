set -x
mkdir -p gpurun_out/r4b
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_native_comm.py -x -q -s -k "not full_size" > gpurun_out/r4b/native.log 2>&1; echo "native rc=$?" >> gpurun_out/r4b/rc.txt
timeout 1200 python -m pytest tests/test_gpu_bench_launcher.py -x -q -s > gpurun_out/r4b/launcher.log 2>&1; echo "launcher rc=$?" >> gpurun_out/r4b/rc.txt
( time timeout 900 python bench.py > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err ) 2> gpurun_out/r4b/bench.time; echo "bench rc=$?" >> gpurun_out/r4b/rc.txt
cat gpurun_out/r4b/rc.txt; tail -5 gpurun_out/r4b/native.log; tail -5 gpurun_out/r4b/launcher.log; cat gpurun_out/r4b/bench.time
