#!/bin/bash
R=$PWD; export TMPDIR=/tmp PYTHONPATH=$R
for n in base "$@"; do so=$R/cleanba_amd/abl_$n.so; [ $n = base ] && so=$R/cleanba_amd/libcleanba_mi.so
 cd /tmp; CBM_SO=$so timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ip_$n -o ip -- python $R/tools/impala_probe.py > /dev/null 2>&1; cd $R
 echo -n "$n: "; python tools/rocprof_summary.py $(find gpurun_out/ip_$n -name "*.db" | head -1) 2>/dev/null | grep -E "tail" | cut -c1-100; rm -rf gpurun_out/ip_$n; done
