cd /root/repo
timeout 900 python -m pytest tests/test_gpu_native_comm.py -x -q -s -k "configs4" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_bench_launcher.py -x -q -s -k "gpus_8" 2>&1 | tail -8
CBM_FORCE_DEVICE=0 timeout 600 python bench.py --topology "2x(a0-l1,2,3)" --env-id Atari57Mix-v5 --steps 4 --warmup 2 > gpurun_out/r4i_configs4.json 2> gpurun_out/r4i_configs4.err; tail -c 900 gpurun_out/r4i_configs4.json
