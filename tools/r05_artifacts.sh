#!/bin/bash
# usage (GPU box, repo root): GIT_REV=<rev> tools/r05_artifacts.sh <tag>   -> gpurun_out/<tag>/*: everything profiles/r05_* is copied from
#   bench_line.json            python bench.py (the driver's command)
#   bench_kernel_stats.md      rocprofv3 --kernel-trace --stats over bench.py; pmc_summary.md / pmc_traffic.json: four separate --pmc passes (Nature minibatch)
#   resnet_kernel_stats.md     rocprofv3 --kernel-trace over one isolated ResNet learner minibatch x 8 + rollout (tools/rn_microbench.py); resnet_pmc_summary.md
#   impala_*                   rocprofv3 --kernel-trace over tools/impala_probe.py (T = 128 fp32 / bf16, T = 20) + the probe's own lines
#   readme_table.txt           tools/readme_table.py (secondary workloads through the product trainer)
#   actor_progress.md          tools/actor_progress.py over a kernel trace of tools/pipeline_probe.py (rollout progress rate inside each learner kernel)
tag=$1; R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$R
t0=$(date +%s); timeout 600 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "python bench.py (no flags): $(( $(date +%s) - t0 )) s wall" > $out/bench_wall.txt
bash tools/pmc_collect.sh $tag > $out/pmc_collect.log 2>&1
mv $out/kernel_stats.md $out/bench_kernel_stats.md
prof() { # name, command...
  local n=$1; shift
  cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $out/tr_$n -o t -- "$@" > $out/$n.log 2>&1; cd $R
  python tools/rocprof_summary.py $(find $out/tr_$n -name "*.db" | head -1) > $out/${n}_kernel_stats.md 2>&1; rm -rf $out/tr_$n
}
prof resnet python $R/tools/rn_microbench.py 8
pmc() { cd /tmp; timeout 400 rocprofv3 --pmc $2 --kernel-trace -d $out/rp_$1 -o p -- python $R/tools/rn_microbench.py 2 > $out/rp_$1.log 2>&1; cd $R; }
pmc sq "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
pmc lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
pmc fetch "FETCH_SIZE"
pmc write "WRITE_SIZE"
db() { find $out/rp_$1 -name "*.db" | head -1; }
python tools/pmc_report.py $(db sq) $(db lds) $(db fetch) $(db write) 2>&1 | sed 's#tools/microbench.py 3 --plain#tools/rn_microbench.py 2#; s#learner kernels of one 3840-frame PPO minibatch, isolated#IMPALA-ResNet: one 128-step rollout + three 3840-frame PPO minibatches#' > $out/resnet_pmc_summary.md
rm -rf $out/rp_sq $out/rp_lds $out/rp_fetch $out/rp_write
prof impala_t128 python $R/tools/impala_probe.py
BF16=1 prof impala_t128_bf16 python $R/tools/impala_probe.py
T=20 prof impala_t20 python $R/tools/impala_probe.py
( python tools/impala_probe.py; BF16=1 python tools/impala_probe.py; T=20 python tools/impala_probe.py; T=20 BF16=1 python tools/impala_probe.py ) 2>&1 | grep -v amdgpu > $out/impala_probe.txt
python tools/readme_table.py 2>&1 | grep -v amdgpu > $out/readme_table.txt
( for t in 1 2 1 2 1 2; do python tools/host_loop_probe.py $t; done ) 2>&1 | grep -v amdgpu > $out/host_loop_probe.txt
NET=resnet python tools/pipeline_probe.py 2>&1 | grep -v amdgpu > $out/resnet_pipeline_probe.txt
python tools/pipeline_probe.py 2>&1 | grep -v amdgpu > $out/nature_pipeline_probe.txt
timeout 200 tools/ubench/gemm2 0 > $out/ubench_gemm2.txt 2>&1
timeout 100 tools/ubench/mfma_shape > $out/ubench_mfma_shape.txt 2>&1
cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $out/tr_ap -o t -- python $R/tools/pipeline_probe.py > /dev/null 2>&1; cd $R
python tools/actor_progress.py $(find $out/tr_ap -name "*.db" | head -1) > $out/actor_progress.md 2>&1; rm -rf $out/tr_ap
# round 4 additions
#   actor_probe.txt / actor_kernel_stats.md   the actor step alone (rollout-only context): us per 120-env step, per-kernel averages
#   tail_trace.txt / heads_trace.txt          in-kernel clock stamps of the per-frame actor tail / the fused PPO heads (needs cleanba_amd/abl_tailtrace.so: tools/variants.sh tailtrace "-DCBM_TAIL_TRACE")
#   il_trace.txt                              phase stamps of impala_loss_kernel (needs cleanba_amd/abl_iltrace.so: tools/variants.sh iltrace "-DCBM_IL_TRACE")
#   learner_only_kernel_stats.md              rocprofv3 --kernel-trace over tools/microbench.py --plain: the twelve GEMMs WITHOUT a concurrent rollout
#   bench_line_a0-l1,2,3_one_gpu.json         BASELINE configs[3] as four role processes on this GPU, native all-reduce
#   bench_line_dp4_one_gpu.json               bench.py --gpus 4 with every rank on this GPU: native 4-rank all-reduce + the configs[3] line as baseline_config
( python tools/actor_probe.py 20; ALGO=impala python tools/actor_probe.py 20; NET=resnet python tools/actor_probe.py 5 ) 2>&1 | grep -v amdgpu > $out/actor_probe.txt
prof actor python $R/tools/actor_probe.py 5
[ -f cleanba_amd/abl_tailtrace.so ] && CBM_SO=$R/cleanba_amd/abl_tailtrace.so python tools/tail_trace.py 2>&1 | grep -v amdgpu > $out/tail_trace.txt
[ -f cleanba_amd/abl_tailtrace.so ] && CBM_SO=$R/cleanba_amd/abl_tailtrace.so python tools/heads_trace.py 2>&1 | grep -v amdgpu > $out/heads_trace.txt
[ -f cleanba_amd/abl_iltrace.so ] && CBM_SO=$R/cleanba_amd/abl_iltrace.so python tools/il_trace.py 2>&1 | grep -v amdgpu > $out/il_trace.txt
prof learner_only python $R/tools/microbench.py 8 --plain
python tools/microbench.py 8 2>&1 | grep -v amdgpu > $out/microbench.txt
CBM_FORCE_DEVICE=0 timeout 300 python bench.py --topology a0-l1,2,3 --steps 6 --warmup 2 > "$out/bench_line_a0-l1,2,3_one_gpu.json" 2>> $out/bench.err
CBM_FORCE_DEVICE=0 timeout 400 python bench.py --gpus 4 --steps 4 --warmup 2 --no-cpu-baseline --no-host-env > $out/bench_line_dp4_one_gpu.json 2>> $out/bench.err
# round 5 additions
#   ipc_stress.txt             tools/ipc_stress.py: 8 processes x 40 create / export / map / all-reduce / teardown cycles, 'safe' (unmap -> barrier -> free, what ships) then
#                              'racy' (round 4's order: the reproducer of the red driver run — expected to FAIL in its second cycle)
#   flaky_loop.txt             tools/flaky_loop.sh: `bench.py --gpus 8` (dp line + configs[4] topology line) FLAKY_ITERS times on this one GPU
#   gpu_pytest.log             the whole `pytest -m gpu` suite at this commit
( python tools/ipc_stress.py 8 40 safe; echo "# racy = round 4's teardown order (free while peers still map): the reproducer, expected to fail"; python tools/ipc_stress.py 8 40 racy ) 2>&1 | grep -v "amdgpu\|c10d\|^frame #\|Cannot find CO" | cut -c1-400 > $out/ipc_stress.txt
AMD_LOG_LEVEL=0 tools/flaky_loop.sh $out/flaky ${FLAKY_ITERS:-8} 8 > /dev/null 2>&1; cp $out/flaky/summary.txt $out/flaky_loop.txt; rm -rf $out/flaky
timeout 1500 python -m pytest tests -m gpu -q > $out/gpu_pytest.log 2>&1
ls $out
