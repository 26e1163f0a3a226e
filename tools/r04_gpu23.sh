cd /root/repo
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "priority" 2>&1 | tail -3
for i in 1 2 3; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-baseline-config 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('value', d['value'], 'ms', d['ms_per_step'], 'host_env', d.get('host_env',{}).get('actor_threads_1'), d.get('host_env',{}).get('actor_threads_2'))"
done
