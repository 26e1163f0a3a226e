cd /root/repo
CBM_SO=$PWD/cleanba_amd/abl_tailtrace.so timeout 120 python tools/heads_trace.py 2>&1 | grep -v amdgpu
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_e2e.py tests/test_gpu_split.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
R=$PWD; cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4k/prof -o a -- python $R/tools/impala_probe.py > $R/gpurun_out/r4k_probe.log 2>&1
cd $R; python tools/rocprof_summary.py $(find gpurun_out/r4k/prof -name "*.db" | head -1) 2>&1 | grep -E "impala_loss|heads_dgrad|total kernel" | cut -c1-120; rm -rf gpurun_out/r4k/prof
timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu | tail -4
