cd /root/repo
python - <<'PY'
import ctypes
h = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so"); lo = ctypes.c_int(); hi = ctypes.c_int()
print("hipDeviceGetStreamPriorityRange rc", h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "lo", lo.value, "hi", hi.value)
PY
for p in none actor actor_hi learner_lo none actor actor_hi learner_lo; do
  if [ $p = none ]; then unset CBM_STREAM_PRIO; else export CBM_STREAM_PRIO=$p; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-baseline-config 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('prio=$p', 'value', d['value'], 'ms', d['ms_per_step'], 'host_env', d.get('host_env',{}).get('actor_threads_1'), d.get('host_env',{}).get('actor_threads_2'))"
done
