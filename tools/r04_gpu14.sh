cd /root/repo
timeout 900 python -m pytest tests/test_gpu_native_comm.py -x -q -k "rank_ordered or dead_peer or small" 2>&1 | tail -3
timeout 600 python tools/native_allreduce_bench.py 2 3 4 8 2>&1 | grep -v amdgpu
