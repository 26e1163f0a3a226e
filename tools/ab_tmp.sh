cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_overlap.py tests/test_gpu_resnet.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for v in prev new prev new; do
  if [ $v = new ]; then unset CBM_SO; else export CBM_SO=$PWD/cleanba_amd/abl_$v.so; fi
  echo "$v: $(timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu | tr '\n' ' ')"
done
