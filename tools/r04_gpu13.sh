cd /root/repo
for v in base c1a c1b c1c c2a c2b c2c c2d c3a c3b c3c da db base; do
so=$PWD/cleanba_amd/abl_$v.so; [ $v = base ] && so=$PWD/cleanba_amd/libcleanba_mi.so
echo -n "$v: "; CBM_SO=$so timeout 120 python tools/actor_probe.py 12 2>&1 | grep -v amdgpu | tail -1
done
