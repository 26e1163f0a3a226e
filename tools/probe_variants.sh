#!/bin/bash
# GPU box: rollout-alone / update-alone / pipelined times (tools/pipeline_probe.py) for the shipped library and ablation variants
for n in base "$@"; do so=cleanba_amd/abl_$n.so; [ $n = base ] && so=cleanba_amd/libcleanba_mi.so; echo "== $n"; CBM_SO=$PWD/$so python tools/pipeline_probe.py 2>&1 | grep -E "rollout alone|pipelined"; done
