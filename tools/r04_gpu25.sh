cd /root/repo
for nb in 256 240 192 256 240 224 160; do
  echo "CBM_PERSIST_BLOCKS=$nb: $(CBM_PERSIST_BLOCKS=$nb timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu | tr '\n' ' ')"
done
for nb in 256 240; do
  echo "IMPALA CBM_PERSIST_BLOCKS=$nb: $(CBM_PERSIST_BLOCKS=$nb timeout 300 python tools/impala_probe.py 2>&1 | grep -v amdgpu | tr '\n' ' ')"
done
