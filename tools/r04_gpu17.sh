cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 300 python tools/microbench.py 16 2>&1 | grep -v amdgpu
timeout 300 python tools/microbench.py 16 2>&1 | grep -v amdgpu | head -1
