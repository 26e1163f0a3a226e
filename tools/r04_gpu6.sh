mkdir -p gpurun_out/r4f
cd /root/repo
CBM_SO=$PWD/cleanba_amd/abl_tailtrace.so timeout 300 python tools/tail_trace.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/actor_probe.py 20 2>&1 | grep -v amdgpu.ids
ALGO=impala timeout 300 python tools/actor_probe.py 20 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py tests/test_gpu_resnet.py -x -q 2>&1 | tail -4
timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu.ids
