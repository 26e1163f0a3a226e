cd /root/repo
for v in c3ns5; do CBM_SO=$PWD/cleanba_amd/abl_$v.so timeout 900 python -m pytest tests/test_gpu_fullsize_oracle.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -2; done
for v in base c3ns5 c3ns6 base c3ns5 c3ns6; do
  if [ $v = base ]; then unset CBM_SO; else export CBM_SO=$PWD/cleanba_amd/abl_$v.so; fi
  echo "$v: $(timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu | tr '\n' ' ')"
done
