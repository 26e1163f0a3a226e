cd /root/repo
CBM_SO=$PWD/cleanba_amd/abl_tailtrace.so python tools/heads_trace.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 300 python tools/microbench.py 16 --plain 2>&1 | grep -v amdgpu
