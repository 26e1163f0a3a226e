cd /root/repo
for v in base skip3 skip23; do
so=$PWD/cleanba_amd/abl_$v.so; [ $v = base ] && so=$PWD/cleanba_amd/libcleanba_mi.so
echo "== $v"; CBM_SO=$so timeout 300 python tools/pipeline_probe.py 2>&1 | grep -v amdgpu.ids
done
