cd /root/repo
CBM_SO=$PWD/cleanba_amd/abl_s16trace.so python tools/s16_trace.py 2>&1 | grep -v amdgpu | grep -E "us since|chunk 7 done|chunk 8 done|chunk 6 done|stored" | cut -c1-40
timeout 120 python tools/actor_probe.py 20 2>&1 | grep -v amdgpu
ALGO=impala timeout 120 python tools/actor_probe.py 20 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_resnet.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu | head -3
