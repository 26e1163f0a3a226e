cd /root/repo
for v in old base old base; do
so=$PWD/cleanba_amd/abl_$v.so; [ $v = base ] && so=$PWD/cleanba_amd/libcleanba_mi.so
echo "== $v"
CBM_SO=$so timeout 300 python tools/host_loop_probe.py 1 2>&1 | grep -v amdgpu.ids | tail -4
CBM_SO=$so timeout 300 python tools/impala_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
done
for p in actor learner; do echo "== prio $p"; CBM_STREAM_PRIO=$p timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu.ids | head -3; done
