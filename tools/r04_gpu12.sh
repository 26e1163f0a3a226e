cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_fullsize.py tests/test_gpu_overlap.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
for ov in 0 1 0 1; do echo "== CBM_BWD_OVERLAP=$ov"; CBM_BWD_OVERLAP=$ov timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu | head -3; CBM_BWD_OVERLAP=$ov timeout 300 python tools/microbench.py 16 --plain 2>&1 | grep -v amdgpu; done
for ov in 0 1; do echo "== impala CBM_BWD_OVERLAP=$ov"; CBM_BWD_OVERLAP=$ov timeout 300 python tools/impala_probe.py 2>&1 | grep -v amdgpu | tail -2; done
