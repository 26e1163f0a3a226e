cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_fullsize.py tests/test_gpu_overlap.py tests/test_gpu_resnet.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
R=$PWD; cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4j/prof -o a -- python $R/tools/microbench.py 8 --plain > /dev/null 2>&1
cd $R; python tools/rocprof_summary.py $(find gpurun_out/r4j/prof -name "*.db" | head -1) 2>&1 | grep -E "reduce|adam|sqnorm|heads" | cut -c1-120; rm -rf gpurun_out/r4j/prof
timeout 300 python tools/microbench.py 16 --plain 2>&1 | grep -v amdgpu
timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu | head -3
