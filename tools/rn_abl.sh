export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/rnabl
for n in libcleanba_mi abl_c1 abl_c2 abl_c4 abl_c8 abl_c14 abl_c15; do
  cd /tmp; CBM_SO=$R/cleanba_amd/$n.so PYTHONPATH=$R timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/rnabl/$n -o rn -- python $R/tools/rn_microbench.py 6 > $R/gpurun_out/rnabl/$n.log 2>&1; cd $R
  python tools/rocprof_summary.py $(find gpurun_out/rnabl/$n -name "*.db" | head -1) > gpurun_out/rnabl/$n.md; rm -rf gpurun_out/rnabl/$n
  tail -1 gpurun_out/rnabl/$n.log
done
python - <<'PY'
import re
names=["libcleanba_mi","abl_c1","abl_c2","abl_c4","abl_c8","abl_c14","abl_c15"]
tab={}
for n in names:
    for l in open(f"gpurun_out/rnabl/{n}.md"):
        p=[x.strip() for x in l.split("|")]
        if len(p)>7 and p[1].startswith("void rn_conv_kernel"):
            tab.setdefault(p[1][19:70],{})[n]=float(p[6])
print("%-52s"%"kernel (max us)"+"".join("%9s"%n[-7:] for n in names))
for k,v in sorted(tab.items()): print("%-52s"%k+"".join("%9.0f"%v.get(n,0) for n in names))
PY
