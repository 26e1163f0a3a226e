#!/bin/bash
# GPU box: rocprofv3 average of the kernels matching $PAT (default "tail") over a probe script ($PROBE, default tools/impala_probe.py), for
# the shipped library and the ablation builds named on the command line (tools/variants.sh)
R=$PWD; export TMPDIR=/tmp PYTHONPATH=$R
PAT=${PAT:-tail}; PROBE=${PROBE:-tools/impala_probe.py}
for n in base "$@"; do so=$R/cleanba_amd/abl_$n.so; [ $n = base ] && so=$R/cleanba_amd/libcleanba_mi.so
 cd /tmp; CBM_SO=$so timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ip_$n -o ip -- python $R/$PROBE $PROBE_ARGS > /dev/null 2>&1; cd $R
 echo -n "$n: "; python tools/rocprof_summary.py $(find gpurun_out/ip_$n -name "*.db" | head -1) 2>/dev/null | grep -E "$PAT" | cut -c1-100; rm -rf gpurun_out/ip_$n; done
