mkdir -p gpurun_out/r4e
cd /root/repo
for v in "$@"; do
CBM_SO=$PWD/cleanba_amd/abl_$v.so timeout 300 python tools/tail_trace.py > gpurun_out/r4e/${v}_trace.log 2>&1
CBM_SO=$PWD/cleanba_amd/abl_$v.so timeout 300 python tools/actor_probe.py 20 > gpurun_out/r4e/${v}_actor.log 2>&1
echo "== $v"; grep -v amdgpu.ids gpurun_out/r4e/${v}_actor.log gpurun_out/r4e/${v}_trace.log
done
