#!/usr/bin/env python
"""bench.py — env-steps/sec of the rollout+update hot path (BASELINE.json metric) on N MI355X.

Workload (BASELINE.json configs[1]): PPO a0-l0, Nature-CNN fp32, local_num_envs=120, num_steps=128,
num_actor_threads=1, 4 epochs x 4 minibatches, A=18, synthetic Breakout-shaped frames from the device env
(frames are rendered into the HBM ring by the env kernel: inputs are resident in HBM, no PCIe in the timed
region).  One "step" = one full update cycle = one 128x120 rollout (actor forward + Gumbel sampling + env
step per env-step) + one learner update (GAE, adv-norm, 16 x (minibatch fwd+bwd, [all-reduce], Adam)).
With N>1 every rank runs its own 120 envs and learner (the reference's a0_l0_dN topology, README.md:103-108)
and gradients are all-reduced per minibatch over RCCL: weak scaling.

The actor rollout k+1 is enqueued on its own HIP stream while update k runs (--concurrency semantics,
ppo:287-304); a single host thread drives both through the C ABI, so nothing but the two device
synchronisations brackets the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cleanba_amd.lib as L  # noqa: E402
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402

E, T, A, EPOCHS, NMB = 120, 128, 18, 4, 4
MB = E * T // NMB
# implicit-GEMM kernels: id -> (name, flops per launch at minibatch MB)  [2*M*N*K]
KERNELS = {
    0: ("conv1_fwd", 2.0 * MB * 400 * 32 * 256), 1: ("conv2_fwd", 2.0 * MB * 81 * 64 * 512), 2: ("conv3_fwd", 2.0 * MB * 49 * 64 * 576),
    3: ("dense_fwd", 2.0 * MB * 512 * 3136), 4: ("heads_wgrad", 2.0 * MB * 512 * 32), 5: ("dense_dgrad", 2.0 * MB * 3136 * 512),
    6: ("dense_wgrad", 2.0 * MB * 3136 * 512), 7: ("conv3_dgrad", 2.0 * MB * 49 * 64 * 576), 8: ("conv3_wgrad", 2.0 * MB * 49 * 576 * 64),
    9: ("conv2_dgrad", 2.0 * MB * 81 * 64 * 512), 10: ("conv2_wgrad", 2.0 * MB * 81 * 512 * 64), 11: ("conv1_wgrad", 2.0 * MB * 400 * 256 * 32),
}
# flops are ALGORITHMIC (a layer's dgrad and wgrad each cost its forward flops).  Both position-major dgrads (conv2, conv3) skip
# the taps that fall into dY's zero border per tile, so they EXECUTE exactly the algorithmic count.
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def cpu_baseline(params, n_frames=768):
    """The CPU restatement (oracle/, 'port') timed on a bounded sample of the same per-env-step work:
    t_fwd (actor forward+sampling) and t_fb (learner forward+loss+backward) per frame on n_frames
    Breakout-shaped frames; env-steps/s = 1 / (t_fwd*(1+1/T) + EPOCHS*t_fb)."""
    import oracle
    cores = min(os.cpu_count() or 1, 32, n_frames // 6)  # OpenMP over frames: more threads than frames/6 only adds reduction cost
    oracle.set_threads(cores)
    st, obs = L.synth_env_reset_host(1, n_frames)
    rng = np.random.default_rng(0)
    for _ in range(4):
        L.synth_env_step_host(1, st, obs, rng.integers(0, A, n_frames).astype(np.int32))
    t0 = time.time()
    logits, value = oracle.nature_forward(params, A, obs, ksplit=14)
    actions, lp, _ = oracle.sample_actions(logits, prng.prng_key(1))
    t_fwd = (time.time() - t0) / n_frames
    adv = rng.normal(size=n_frames).astype(np.float32)
    t0 = time.time()
    oracle.ppo_loss_grad(params, A, obs, None, actions, lp, adv, value + adv)
    t_fb = (time.time() - t0) / n_frames
    sps = 1.0 / (t_fwd * (1.0 + 1.0 / T) + EPOCHS * t_fb)
    # SURVEY section 8d row (A), "reference-faithful threading": the reference pins XLA-CPU to one intra-op thread per computation
    # (ppo:28), i.e. one busy core for the actor thread and one for the learner, running concurrently -> the slower of the two bounds it
    oracle.set_threads(1)
    n1 = 24
    t0 = time.time()
    lg1, v1 = oracle.nature_forward(params, A, obs[:n1], ksplit=14)
    a1, lp1, _ = oracle.sample_actions(lg1, prng.prng_key(1))
    t_fwd1 = (time.time() - t0) / n1
    t0 = time.time()
    oracle.ppo_loss_grad(params, A, obs[:n1], None, a1, lp1, adv[:n1], v1 + adv[:n1])
    t_fb1 = (time.time() - t0) / n1
    sps_ref_threads = 1.0 / max(t_fwd1 * (1.0 + 1.0 / T), EPOCHS * t_fb1)
    return {"value": round(sps, 2), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "reference_threading": {"value": round(sps_ref_threads, 2), "cores": 2,
                                    "note": f"one thread per role like the reference (ppo:28): actor {t_fwd1 * 1e3:.1f} ms/frame and learner "
                                            f"{t_fb1 * 1e3:.1f} ms/frame on one core each, overlapped; {n1}-frame sample"},
            "sample": f"{n_frames} synthetic frames: oracle actor forward+sampling ({t_fwd * 1e3:.2f} ms/frame) and PPO "
                      f"forward+loss+backward ({t_fb * 1e3:.2f} ms/frame) with OpenMP over frames; per-env-step cost = "
                      f"t_fwd*(1+1/{T}) + {EPOCHS}*t_fb (Adam/GAE/shuffle excluded: <1%)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prof-kernel", type=int, default=9, help="igemm kernel id timed with HIP events for the roofline line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bwd-split", type=int, default=0, choices=[0, 2, 3],
                    help="build-only extension (NOT the headline): backward GEMMs on split-bf16 MFMA, see cbm_config.backward_split")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    force_dist = os.environ.get("CBM_FORCE_DIST") == "1"  # exercise the N>1 code path (split form + all-reduce) on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                init_method=f"tcp://{os.environ.get('MASTER_ADDR', '127.0.0.1')}:{os.environ.get('MASTER_PORT', '29500')}")
    cfg = L.default_config(L.ALGO_PPO)
    cfg.device = local_rank
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions = E, 1, T, A
    cfg.backward_split = a.bwd_split
    from cleanba_amd.trainer import HipEngine
    ctx = HipEngine(cfg)
    key = prng.prng_key(1)
    key, nk, ak, ck = prng.split(key, 4)
    params = M.init_nature_params(A, nk, ak, ck)
    ctx.set_params(params)
    ctx.actor_set_key(0, key)
    ctx.actor_env_reset_device(0, 1 + rank)  # env seed = seed + process_index + thread id (ppo:238)
    lkey = key.copy()
    from cleanba_amd.trainer import GradAllReducer
    allreduce = GradAllReducer(ctx, world, dist_module=dist, active=dist is not None)   # tail of the gradient overlaps the conv backward
    n_opt = EPOCHS * NMB
    total_updates = a.warmup + a.steps
    opt_count = 0

    def rollout():
        ctx.actor_begin_rollout(0, True)
        ctx.actor_rollout_device(0, T)
        ctx.actor_commit(0)

    def update(v):
        nonlocal lkey, opt_count
        ctx.learner_wait()
        lrs = [M.linear_schedule(opt_count + i, 2.5e-4, n_opt, max(total_updates, 1)) for i in range(n_opt)]
        bcs = [M.adam_bias_corrections(opt_count + i + 1) for i in range(n_opt)]
        if dist is None:
            lkey, _ = ctx.learner_update(lkey, lrs, [b[0] for b in bcs], [b[1] for b in bcs], want_stats=False)
        else:
            lkey = ctx.learner_prepare(lkey)
            i = 0
            for e in range(EPOCHS):
                lkey = ctx.learner_epoch_begin(lkey)
                for mb in range(NMB):
                    ctx.learner_minibatch_grad(e, mb)
                    grad_div = allreduce()
                    ctx.learner_optimizer_step(float(lrs[i]), float(bcs[i][0]), float(bcs[i][1]), grad_div)
                    i += 1
            ctx.learner_finish(n_opt, want_stats=False)
        opt_count += n_opt

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            ctx.sync()

    rollout()  # rollout 1 (params v0)
    for v in range(1, a.warmup + 1):
        rollout()  # rollout v+1 overlaps update v
        update(v)
    barrier()
    if a.prof_kernel >= 0:
        ctx.profile_select(a.prof_kernel)
    t0 = time.perf_counter()
    for v in range(a.warmup + 1, total_updates + 1):
        rollout()
        update(v)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    prof_ms, prof_n = ctx.profile_read() if a.prof_kernel >= 0 else (0.0, 0)
    _, stats = None, None
    env_steps = a.steps * T * E * world
    sps = env_steps / dt

    if rank == 0:
        kname, kflops = KERNELS.get(a.prof_kernel, ("none", 0.0))
        roof = None
        if prof_n > 0:
            avg_s = prof_ms / prof_n / 1e3
            ach = kflops / avg_s / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get(str(a.prof_kernel), {}).get("traffic_bytes")
            roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "launches": prof_n, "avg_us": round(avg_s * 1e6, 1),
                    "flops_per_launch": kflops}
        line = {"metric": "env-steps/sec (whole node), Breakout-v5 84x84x4, num_envs=120", "value": round(sps, 1), "unit": "env-steps/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None,
                "dtype": "f32" if not a.bwd_split else f"f32 forward / split-bf16 x{a.bwd_split} backward GEMMs (extension, not the headline)", "data": "synthetic",
                "config": {"workload": "PPO a0-l0-d%d: Nature-CNN fp32, local_num_envs=120, rollout_len=128, 4 epochs x 4 minibatches, A=18, "
                                       "device synthetic Breakout-shaped env, concurrency on" % world,
                           "global_batch": T * E * world, "parallelism": f"dp{world}"},
                "roofline": roof}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(params)
        _emit(json.dumps(line))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


_REAL_STDOUT = None


def _emit(text):
    """The one JSON line goes to the process's ORIGINAL stdout; see _quiet_stdout."""
    if _REAL_STDOUT is None:
        print(text, flush=True)
    else:
        os.write(_REAL_STDOUT, (text + "\n").encode())


def _quiet_stdout():
    """Everything else that writes to fd 1 — RCCL's start-up banner (C stdio, flushed at exit, i.e. AFTER the result line), library warnings,
    the other ranks under torchrun — is sent to stderr, so stdout carries exactly one line: rank 0's JSON."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


if __name__ == "__main__":
    _quiet_stdout()
    main()
