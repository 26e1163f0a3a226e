set -x
mkdir -p gpurun_out/r4a
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_native_comm.py -x -q -s -k "not full_size" > gpurun_out/r4a/native.log 2>&1; echo "native rc=$?" >> gpurun_out/r4a/rc.txt
timeout 1500 python -m pytest tests/test_gpu_native_comm.py -x -q -s -k "full_size" > gpurun_out/r4a/native_full.log 2>&1; echo "native_full rc=$?" >> gpurun_out/r4a/rc.txt
timeout 900 python -m pytest tests/test_gpu_resnet_hidden.py -x -q -s > gpurun_out/r4a/hidden.log 2>&1; echo "hidden rc=$?" >> gpurun_out/r4a/rc.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -s > gpurun_out/r4a/fullsize.log 2>&1; echo "fullsize rc=$?" >> gpurun_out/r4a/rc.txt
timeout 300 python tools/pipeline_probe.py > gpurun_out/r4a/probe.log 2>&1
cat gpurun_out/r4a/rc.txt; tail -5 gpurun_out/r4a/native.log; tail -3 gpurun_out/r4a/probe.log
