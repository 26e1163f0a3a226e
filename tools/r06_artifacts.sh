#!/bin/bash
# usage (GPU box, repo root; the caller writes `git rev-parse --short HEAD` into .build_rev first — the snapshot has no .git):
#   tools/r06_artifacts.sh <tag>      -> gpurun_out/<tag>/*: everything profiles/r06_* is copied from (at ONE library revision, recorded in every file that has a place for it)
#   bench_line.json / bench_wall.txt   python bench.py (the driver's command) and its wall time
#   bench_kernel_stats.md, pmc_summary.md, pmc_traffic.json   tools/pmc_collect.sh: rocprofv3 --kernel-trace --stats over bench.py's headline leg; four separate --pmc passes
#   microbench.txt, learner_only_kernel_stats.md              the twelve GEMMs of one 3840-frame minibatch, isolated (HIP events / kernel trace)
#   nature_pipeline_probe.txt, resnet_pipeline_probe.txt      rollout alone / update alone / pipelined
#   actor_probe.txt, actor_kernel_stats.md                    the actor step alone (five launches), CBM_ACTOR_FUSED=1 beside it
#   impala_probe.txt, impala_*_kernel_stats.md, readme_table.txt, host_loop_probe.txt
#   resnet_kernel_stats.md, resnet_roofline.md, resnet_pmc_summary.md
#   bench_line_dp2_one_gpu.json, bench_line_dp4_one_gpu.json, bench_line_a0-l1,2,3_one_gpu.json   the N > 1 code on ONE GPU (CBM_FORCE_DEVICE=0): native all-reduce, allreduce_ab with the overlap probe
#   ipc_stress.txt, gpu_pytest.log
tag=$1; R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$R GIT_REV=$(cat $R/.build_rev 2>/dev/null || echo unknown)
echo "library revision: $GIT_REV" > $out/REVISION.txt
# counters first: bench.py reports a kernel's `traffic` from profiles/pmc_traffic.json and only for the kernel symbols recorded there, so the table of THIS
# library revision has to be in place before the line is taken (a kernel renamed since the last collection would read null)
bash tools/pmc_collect.sh $tag > $out/pmc_collect.log 2>&1
mv $out/kernel_stats.md $out/bench_kernel_stats.md
python -c "import json,sys; d=json.load(open(sys.argv[1])); assert len(d) >= 12, len(d)" $out/pmc_traffic.json && cp $out/pmc_traffic.json profiles/pmc_traffic.json
t0=$(date +%s); timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "python bench.py (no flags): $(( $(date +%s) - t0 )) s wall" > $out/bench_wall.txt
prof() { # name, command...
  local n=$1; shift
  cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $out/tr_$n -o t -- "$@" > $out/$n.log 2>&1; cd $R
  python tools/rocprof_summary.py $(find $out/tr_$n -name "*.db" | head -1) > $out/${n}_kernel_stats.md 2>&1; rm -rf $out/tr_$n
}
python tools/microbench.py 8 2>&1 | grep -v amdgpu > $out/microbench.txt
prof learner_only python $R/tools/microbench.py 8 --plain
python tools/pipeline_probe.py 2>&1 | grep -v amdgpu > $out/nature_pipeline_probe.txt
NET=resnet python tools/pipeline_probe.py 2>&1 | grep -v amdgpu > $out/resnet_pipeline_probe.txt
( python tools/actor_probe.py 20; ALGO=impala python tools/actor_probe.py 20; NET=resnet python tools/actor_probe.py 5; echo "# CBM_ACTOR_FUSED=1 (the dataflow experiment)"; CBM_ACTOR_FUSED=1 python tools/actor_probe.py 20 ) 2>&1 | grep -v amdgpu > $out/actor_probe.txt
prof actor python $R/tools/actor_probe.py 5
prof resnet python $R/tools/rn_microbench.py 8
python tools/resnet_roofline.py $out/resnet_kernel_stats.md > $out/resnet_roofline.md 2>&1
pmc() { cd /tmp; timeout 400 rocprofv3 --pmc $2 --kernel-trace -d $out/rp_$1 -o p -- python $R/tools/rn_microbench.py 2 > $out/rp_$1.log 2>&1; cd $R; }
pmc sq "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
pmc lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
pmc fetch "FETCH_SIZE"
pmc write "WRITE_SIZE"
db() { find $out/rp_$1 -name "*.db" | head -1; }
python tools/pmc_report.py $(db sq) $(db lds) $(db fetch) $(db write) 2>&1 | sed 's#tools/microbench.py 3 --plain#tools/rn_microbench.py 2#; s#learner kernels of one 3840-frame PPO minibatch, isolated#IMPALA-ResNet: one 128-step rollout + three 3840-frame PPO minibatches#' > $out/resnet_pmc_summary.md
rm -rf $out/rp_sq $out/rp_lds $out/rp_fetch $out/rp_write $out/rp_*.log
prof impala_t128 python $R/tools/impala_probe.py
BF16=1 prof impala_t128_bf16 python $R/tools/impala_probe.py
T=20 prof impala_t20 python $R/tools/impala_probe.py
( python tools/impala_probe.py; BF16=1 python tools/impala_probe.py; T=20 python tools/impala_probe.py; T=20 BF16=1 python tools/impala_probe.py ) 2>&1 | grep -v amdgpu > $out/impala_probe.txt
python tools/readme_table.py 2>&1 | grep -v amdgpu > $out/readme_table.txt
( for t in 1 2 1 2 1 2; do python tools/host_loop_probe.py $t; done ) 2>&1 | grep -v amdgpu > $out/host_loop_probe.txt
CBM_FORCE_DEVICE=0 timeout 400 python bench.py --gpus 2 --steps 4 --warmup 2 --no-baseline-config > $out/bench_line_dp2_one_gpu.json 2>> $out/bench.err
CBM_FORCE_DEVICE=0 timeout 600 python bench.py --gpus 4 --steps 4 --warmup 2 --no-cpu-baseline --no-host-env > $out/bench_line_dp4_one_gpu.json 2>> $out/bench.err
CBM_FORCE_DEVICE=0 timeout 400 python bench.py --topology a0-l1,2,3 --steps 6 --warmup 2 > "$out/bench_line_a0-l1,2,3_one_gpu.json" 2>> $out/bench.err
( python tools/ipc_stress.py 8 20 safe ) 2>&1 | grep -v "amdgpu\|c10d\|^frame #\|Cannot find CO" | cut -c1-400 > $out/ipc_stress.txt
timeout 2400 python -m pytest tests -m gpu -q > $out/gpu_pytest.log 2>&1
tail -3 $out/gpu_pytest.log
ls $out
