#!/usr/bin/env python
"""bench.py — env-steps/sec of the rollout+update hot path (BASELINE.json metric) on N MI355X.

Workload (BASELINE.json configs[1]): PPO a0-l0, Nature-CNN fp32, local_num_envs=120, num_steps=128, num_actor_threads=1,
4 epochs x 4 minibatches, A=18, synthetic Breakout-shaped frames from the device env (frames are rendered into the HBM ring by the env
kernel: inputs are resident in HBM, no PCIe in the timed region).  One "step" = one full update cycle = one 128x120 rollout (actor forward +
Gumbel sampling + env step per env-step) + one learner update (GAE, adv-norm, 16 x (minibatch fwd+bwd, [all-reduce], Adam)).

  python bench.py --gpus N                      N ranks of the reference's a0_l0_dN topology (README.md:103-108): every rank runs its own
                                                120 envs and learner, gradients are all-reduced per minibatch over RCCL (csrc/comm.hip): weak
                                                scaling.  Without torchrun's variables the script launches its N ranks itself.
  python bench.py --gpus 4 --topology a0-l1,2,3               BASELINE configs[3]: 1 actor GPU + 3 learner GPUs (README.md:62)
  python bench.py --gpus 8 --topology "2x(a0-l1,2,3)" --env-id Atari57Mix-v5     configs[4]: two such groups (benchmark.sh:80)
  torchrun --nproc-per-node N bench.py --gpus N ...           the driver's form; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are honoured

The actor rollout k+1 is enqueued on its own HIP stream while update k runs (--concurrency semantics, ppo:287-304); the host is the
trainer's: one thread per actor slot enqueues rollouts, the main thread drives the learner, both through the C ABI, and nothing but the two
device synchronisations brackets the timed region.
"""
import argparse
import json
import os
import re
import subprocess
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
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E ~8 TB/s
ALG_FLOPS_PER_ENV_STEP = 243.2e6   # SURVEY §8d (PPO, Nature-CNN, A=18): rollout forward + 4 epochs x (forward + "2 x forward" backward)
ALG_BYTES_PER_ENV_STEP = 345.8e3
# EXECUTED flops per env-step: SURVEY's "backward = 2 x forward" prices a conv1 INPUT gradient that nobody computes (frames need no gradient:
# not XLA, not this build).  What is launched: the rollout's forward + 4 epochs x the twelve GEMMs of KERNELS (+ the A+1-wide heads forward).
FWD_FLOPS_PER_FRAME = 2.0 * (400 * 32 * 256 + 81 * 64 * 512 + 49 * 64 * 576 + 512 * 3136 + 512 * (A + 1))
EXEC_FLOPS_PER_ENV_STEP = FWD_FLOPS_PER_FRAME + EPOCHS * (sum(f for _, f in KERNELS.values()) / MB + 2.0 * 512 * (A + 1))
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
# conv1 on the bf16 matrix cores (cbm_config.conv1_fp32_chain = 0) is HBM-bound; ALGORITHMIC bytes per launch at minibatch MB:
#   forward: frames in (28,224 B) + fp32 activations out (400 x 32 x 4) + ReLU words (400 x 4); weight gradient: frames + dY in (its 32 KB partial per block is overhead)
CONV1_EXACT_BYTES = {0: MB * (28224 + 400 * 32 * 4 + 400 * 4), 11: MB * (28224 + 400 * 32 * 4)}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


# --------------------------------------------------------------------------------------------- CPU baseline (same harness)
def _cpu_budget():
    """CPUs this process may actually use: os.cpu_count() capped by the cgroup's CPU quota (the GPU boxes of this pool show 256 cores and
    a cpu.max of 16 CPUs: 64 OpenMP threads there are 64 threads time-sliced onto 16)."""
    n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:  # noqa: BLE001
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:  # noqa: BLE001
            pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:  # noqa: BLE001
        pass
    return n, quota


def cpu_baseline(n_envs=16, n_steps=32, updates=3, full_updates=3):
    """The CPU restatement (oracle/, kind "port") driven by the SAME host program as the GPU run — cleanba_amd.trainer.train with the
    oracle-backed engine (tests/oracle_engine.py) in place of the HIP library: same actor thread, same ring hand-off, same counters — on a
    bounded sample of the workload: Nature-CNN PPO, 4 epochs x 4 minibatches, A=18, host synthetic env, `n_envs` envs x `n_steps` steps per
    rollout (the full 120 x 128 rollout would take ~2 minutes per update on these cores), `updates` updates, env-steps/s = the MEDIAN over
    updates of local_batch_size / (time between consecutive update completions) — three full-size intervals for `value`.  Two rows: the oracle's OpenMP over frames on all cores it
    can use, and one intra-op thread per role like the reference pins XLA-CPU (ppo:28)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from oracle_engine import OracleEngine
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train

    def run(threads, n_envs=n_envs, n_steps=n_steps, updates=updates):
        oracle.set_threads(threads)
        stamps = []
        argv = ["--local-num-envs", str(n_envs), "--num-actor-threads", "1", "--num-steps", str(n_steps), "--env-backend", "host", "--network", "nature",
                "--total-timesteps", str((updates + 1) * n_envs * n_steps), "--log-frequency", "100000", "--concurrency",
                "--conv1-fp32-chain", "3"]     # the CPU restatement of the fp32 fmaf chain — not the oracle's emulation of the bf16 matrix instruction, which is a checker's tool
        cwd = os.getcwd()
        os.chdir(os.environ.get("TMPDIR", "/tmp"))
        try:
            with open(os.devnull, "w") as dn:
                so = sys.stdout
                sys.stdout = dn
                try:
                    train(parse_args(argv, "ppo"), "ppo", engine_factory=OracleEngine, on_update=lambda v, s, e: stamps.append(time.perf_counter()))
                finally:
                    sys.stdout = so
        finally:
            os.chdir(cwd)
        dts = np.diff(np.array(stamps))          # the first update's completion is the start mark (warm-up: page-in, thread start)
        return float(n_envs * n_steps / np.median(dts)), [round(float(x), 3) for x in dts]

    ncpu_seen, quota = _cpu_budget()
    ncpu = max(1, min(ncpu_seen, int(quota + 0.5))) if quota else ncpu_seen   # threads beyond the quota only time-slice
    # value: the benchmark's OWN configuration (E = 120 envs x T = 128 steps, 3840-frame minibatches) on all the cores the oracle's OpenMP-over-frames
    # can use, `full_updates` timed update interval(s) after one warm-up update (~10-20 s each)
    cores_full = max(1, min(ncpu, 64))
    sps_full, dts_full = run(cores_full, E, T, full_updates)
    # reduced sample (16 envs x 32 steps, 128-frame minibatches) for the one-thread-per-role row, which would take minutes per update at full size
    cores = max(1, min(ncpu, 32, (n_envs * n_steps // NMB) // 6))   # OpenMP over frames: more threads than frames/6 only adds reduction cost
    sps, dts = run(cores)
    sps1, dts1 = run(1)
    # full-size estimate of the one-thread-per-role row: one thread's time for a PPO minibatch (forward + loss + backward) is linear in the
    # frames, so a 192-frame slice of a 3840-frame minibatch is timed on ONE thread and scaled; the actor's share is the 120-frame forward.
    oracle.set_threads(1)
    rng = np.random.default_rng(7)
    nf = 192
    P1 = M.init_nature_params(A, *prng.split(prng.prng_key(1), 4)[1:])
    fr = rng.integers(0, 256, (nf, 4, 84, 84), dtype=np.uint8)
    acts_, olp, adv_, tgt_ = rng.integers(0, A, nf).astype(np.int32), np.full(nf, -np.log(A), np.float32), rng.normal(size=nf).astype(np.float32), \
        rng.normal(size=nf).astype(np.float32)
    oracle.ppo_loss_grad(P1, A, fr[:8], None, acts_[:8], olp[:8], adv_[:8], tgt_[:8])          # warm the code / pages
    t0 = time.perf_counter()
    oracle.ppo_loss_grad(P1, A, fr, None, acts_, olp, adv_, tgt_)
    t_mb = (time.perf_counter() - t0) * MB / nf                                                   # one 3840-frame minibatch, one thread
    t0 = time.perf_counter()
    oracle.nature_forward(P1, A, fr[:E], ksplit=14)
    t_act = time.perf_counter() - t0                                                              # one 120-env actor forward, one thread
    t_update, t_rollout = EPOCHS * NMB * t_mb, T * t_act
    sps1_full = E * T / max(t_update, t_rollout)                                                   # --concurrency: the two threads overlap
    oracle.set_threads(cores_full)
    return {"value": round(sps_full, 2), "unit": "env-steps/s", "cores": cores_full + 1, "kind": "port", "estimate": False,
            "sample": f"the benchmark's configuration itself: same harness as the GPU run (cleanba_amd.trainer.train, actor thread + learner thread, "
                      f"--concurrency) on the oracle engine, PPO Nature-CNN fp32, {E} envs x {T} steps per rollout, 4 epochs x 4 minibatches of {MB} frames, "
                      f"host synthetic env; {full_updates} timed update interval(s) {dts_full} s after one warm-up update, {cores_full} OpenMP threads in the "
                      f"learner (+1 actor thread); the host shows {ncpu_seen} cores" + (f" under a cgroup quota of {quota:g} CPUs" if quota else "") +
                      ".  A timing of the C restatement, not of JAX.", "host_cpus_visible": ncpu_seen, "host_cpu_quota": quota,
            "reduced_sample": {"value": round(sps, 2), "cores": cores + 1, "estimate": True,
                               "note": f"{n_envs} envs x {n_steps} steps per rollout (128-frame minibatches cap OpenMP at {cores} threads): median of {updates} "
                                       f"update intervals {dts} s — small batches depress CPU efficiency; NOT the benchmark's configuration"},
            "reference_threading_full_size": {"value": round(sps1_full, 2), "cores": 2, "estimate": True,
                                              "note": f"ESTIMATE for the benchmark's own configuration with one intra-op thread per role (ppo:28): a {nf}-frame PPO "
                                                      f"minibatch (forward + loss + backward) timed on one thread and scaled by {MB}/{nf} = {t_mb:.1f} s per "
                                                      f"3840-frame minibatch x 16 = {t_update:.0f} s per update; the actor thread's 128 x 120-frame forwards "
                                                      f"{t_rollout:.1f} s overlap it"},
            "reference_threading": {"value": round(sps1, 2), "cores": 2, "estimate": True,
                                    "note": f"one intra-op thread per role like the reference pins XLA-CPU (ppo:28): one actor thread + one learner thread, on the "
                                            f"reduced sample ({n_envs} x {n_steps}); intervals {dts1} s"}}


# --------------------------------------------------------------------------------------------- data-parallel bench (a0_l0_dN)
def _dry_engine():
    """--dry-run: the CPU stand-in for HipEngine (tests/dry_engine.py: the oracle-backed engine of the CPU tests + gloo).  Exists so that the exact
    launcher / rendezvous / communicator bring-up / timing protocol / watchdog / JSON-merge code of the N = 2 / 4 / 8 lines runs in `pytest -m "not gpu"`
    (tests/test_host_cpu.py) — the first multi-GPU run must not be the first run of that code.  Never a measurement: the line says "dry_run": true."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dry_engine import DryRunEngine
    return DryRunEngine


def run_dp(a, world, rank, local_rank):
    from cleanba_amd import topology
    if a.dry_run:
        HipEngine = _dry_engine()
    else:
        from cleanba_amd.trainer import HipEngine
    cfg = L.default_config(L.ALGO_PPO)
    cfg.device = int(os.environ.get("CBM_FORCE_DEVICE", local_rank))
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps, cfg.num_actions = E, 1, T, A
    cfg.backward_split = a.bwd_split
    cfg.conv1_fp32_chain = a.conv1_fp32_chain
    # rendezvous (imports torch for the TCPStore) BEFORE the first HIP context of the process, like cleanba_amd.trainer: importing torch into a process whose
    # HIP runtime is already live has stalled inside torch's extension load (tests/native_comm_worker.py, round 6)
    rdv = None
    if world > 1:
        rdv = topology.Rendezvous(world, rank, os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "29500"))
    ctx = HipEngine(cfg)
    if world > 1 or ctx.wants_comm_at_world_one():
        topology.setup_learner_comm(ctx, rdv, list(range(world)), rank)      # RCCL communicator over all ranks (pmap's device list, ppo:656-660)
    comm = ctx.comm_size() > 0
    # how many DISTINCT devices the ranks of the communicator sit on (a one-hot of this rank's device ordinal, max over ranks): N for a real N-GPU
    # run, 1 for the CBM_FORCE_DEVICE test shape — so that a line can never pass several ranks time-slicing one GPU off as a multi-GPU number
    distinct_devices = None
    if comm:
        onehot = np.zeros(16)
        onehot[int(cfg.device) % 16] = 1.0
        distinct_devices = int(round(float(np.sum(ctx.comm_allreduce_f64(onehot, "max")))))
    key = prng.prng_key(1)
    key, nk, ak, ck = prng.split(key, 4)
    params = M.init_nature_params(A, nk, ak, ck)
    ctx.set_params(params)
    ctx.actor_set_key(0, key)
    ctx.actor_env_reset_device(0, 1 + rank, a.env_id.startswith("Atari57"))  # env seed = seed + process_index + thread id (ppo:238)
    lkey = key.copy()
    n_opt = EPOCHS * NMB
    total_updates = a.warmup + a.steps
    opt_count = 0

    def rollout():
        ctx.actor_begin_rollout(0, True)
        ctx.actor_rollout_device(0, T)
        ctx.actor_commit(0)

    def update():
        nonlocal lkey, opt_count
        ctx.learner_wait()
        lrs = [M.linear_schedule(opt_count + i, 2.5e-4, n_opt, max(total_updates, 1)) for i in range(n_opt)]
        bcs = [M.adam_bias_corrections(opt_count + i + 1) for i in range(n_opt)]
        # one C call per update; with a communicator it all-reduces every minibatch's gradient (tail under the conv backward) itself
        lkey, _ = ctx.learner_update(lkey, lrs, [b[0] for b in bcs], [b[1] for b in bcs], want_stats=False)
        opt_count += n_opt

    def barrier():
        ctx.sync()
        if comm:
            ctx.comm_barrier()
            ctx.sync()

    # Host structure = the trainer's (cleanba_amd.trainer): one host thread per actor slot enqueues rollouts, the main thread drives the learner;
    # cbm_actor_begin_rollout blocks the actor thread until its ring entry is free and the parameter version it needs exists (ppo:287-304).
    # Timed region: K rollouts + K updates, exactly as many of each as a single-threaded "rollout(); update()" loop would enqueue.
    import threading
    go, failed, committed = threading.Event(), [], [0]

    def actor_thread():
        try:
            for _ in range(a.warmup + 1):      # rollouts 1 .. W+1 (rollout v+1 overlaps update v)
                rollout()
                committed[0] += 1
            go.wait()
            for _ in range(a.steps):           # rollouts W+2 .. W+K+1, inside the timed region
                rollout()
        except BaseException as e:  # noqa: BLE001
            failed.append(e)
            ctx.abort()
            raise

    th = threading.Thread(target=actor_thread, daemon=True)
    th.start()
    for _ in range(a.warmup):
        update()
    while th.is_alive() and not failed and committed[0] < a.warmup + 1:
        time.sleep(1e-4)                       # rollout W+1 has been enqueued; barrier() below waits for it on the device
    barrier()
    if a.prof_kernel != -1:
        ctx.profile_select(a.prof_kernel)
    if comm:
        ctx.comm_profile(True)
    t0 = time.perf_counter()
    go.set()
    prof_steps = a.steps if a.prof_kernel == -1 else max(1, min(a.steps, a.prof_steps if a.prof_steps > 0 else (a.steps + 7) // 8))
    for i in range(a.steps):
        if i == prof_steps and a.prof_kernel != -1:
            ctx.profile_select(-3)             # CBM_PROFILE_PAUSE: the event pairs cost ~6 us each (they serialise back-to-back launches),
        update()                               # so only the first prof_steps steps of the timed region carry them
    th.join()
    if failed:
        raise failed[0]
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    if comm:
        dt = float(ctx.comm_allreduce_f64([dt], "max")[0])
    env_steps = a.steps * T * E * world
    sps = env_steps / dt

    line = None
    if rank == 0:
        line = {"metric": "env-steps/sec (whole node), Breakout-v5 84x84x4, num_envs=120", "value": round(sps, 1), "unit": "env-steps/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None,
                "dtype": "f32" if not a.bwd_split else f"f32 forward / split-bf16 x{a.bwd_split} backward GEMMs (extension, not the headline)", "data": "synthetic",
                "config": {"workload": "PPO a0-l0-d%d: Nature-CNN fp32, local_num_envs=%d, rollout_len=%d, 4 epochs x 4 minibatches, A=18, "
                                       "device synthetic %s env, concurrency on" % (world, E, T, "Atari-57-mix" if a.env_id.startswith("Atari57") else "Breakout-shaped"),
                           "global_batch": T * E * world, "parallelism": f"dp{world}",
                           "conv1_fp32_chain": a.conv1_fp32_chain,
                           "conv1_learner": ("fp32 throughout.  conv1 of the learner's minibatches forms its products EXACTLY on the bf16 matrix cores: a uint8 pixel is exact in "
                                             "bf16, the fp32 operand (w/255 forward, dY weight gradient) is split into three 8-bit terms that sum to it exactly, fp32 accumulate "
                                             "(no operand bit dropped; logits within 1e-6 of the fp32 fmaf chain, bar 1e-5: tests/test_gpu_conv1_exact.py).  The same run on the "
                                             "fp32-MFMA chain kernels (rounds 1-5, bit-identical to the oracle) is secondary.ppo_nature_conv1_fp32_chain")
                                            if not (a.conv1_fp32_chain & 3) else "fp32-MFMA fmaf chains (bit 0: forward, bit 1: weight gradient), bit-identical to the oracle"},
                "per_gpu": {"value": round(sps / world, 1), "rank0_local_ms_per_step": round(dt_local / a.steps * 1e3, 3)},
                "roofline": roofline(ctx, a, dt, prof_steps)}
        if a.dry_run:
            line["dry_run"] = True
            line["data"] = "DRY RUN on the CPU oracle engine over gloo at %d envs x %d steps: plumbing only, not a measurement" % (E, T)
        if comm:
            line["distinct_devices"] = distinct_devices
            tail_ms, exposed_ms, n = ctx.comm_profile_read()
            P = ctx.P
            be = ctx.comm_backend()
            line["allreduce"] = {"backend": {"loopback": "loopback self-test communicator (CBM_COMM_LOOPBACK, not RCCL)", "native": "native",
                                             "rccl": "rccl"}.get(be, be),
                                 "ranks": world, "bytes_per_minibatch": int(P * 4), "minibatches": n,
                                 "tail_bytes": int((P - ctx.grad_tail_offset()) * 4),
                                 "tail_allreduce_us_avg": round(tail_ms / max(n, 1) * 1e3, 1),
                                 "exposed_us_avg": round(exposed_ms / max(n, 1) * 1e3, 1),
                                 "note": "tail (dense + heads) runs on the communication stream under the conv backward; exposed = end of the "
                                         "backward pass -> optimizer may start on the learner stream (head all-reduce + whatever of the tail was not hidden)"}
    if comm and world > 1 and not a.no_allreduce_ab and not a.dry_run:
        try:
            ab = allreduce_ab(ctx, rdv, world, rank)
        except BaseException as e:  # noqa: BLE001
            ab = {"error": f"{type(e).__name__}: {e}"}
        if line is not None:
            line["allreduce_ab"] = ab
    # teardown of a run whose ranks map each other's buffers (native all-reduce): unmap -> barrier -> free.  An owner that frees a window a peer
    # still maps leaves the exporting process's IPC state broken for its NEXT export (tools/ipc_stress.py racy: hipIpcGetMemHandle "invalid
    # argument" / the peers' hipIpcOpenMemHandle "invalid device pointer" one cycle later) — the round-4 failure of the topology phase below.
    ctx.unmap_peers()
    if rdv is not None:
        rdv.barrier("unmapped")
    ctx.close()
    return line, params


def roofline(ctx, a, dt, prof_steps):
    """Per-kernel HIP-event times of the timed region (events bracket every learner-stream launch of ids 0..11).  The `roofline` object is
    the kernel with the LARGEST measured time share; `kernels` lists all twelve; `whole_step` prices the whole step's algorithmic flops."""
    if a.prof_kernel == -1:
        return None
    traffic_tab = json.load(open(TRAFFIC_FILE)) if os.path.exists(TRAFFIC_FILE) else {}
    if a.prof_kernel == -2:
        ms, cnt = ctx.profile_read_all(12)
    else:
        tot, n = ctx.profile_read()
        ms, cnt = np.zeros(12), np.zeros(12, np.int32)
        ms[a.prof_kernel], cnt[a.prof_kernel] = tot, n
    rows = {}
    for k, (name, flops) in KERNELS.items():
        if cnt[k] == 0:
            continue
        avg_s = ms[k] / cnt[k] / 1e3
        tf = flops / avg_s / 1e12
        launched = ctx.profile_kernel_name(k)                 # "<kernel symbol> <functor type>" of what the library launched for this id
        ent = traffic_tab.get(str(k), {})
        tr = ent.get("traffic_bytes") if _same_kernel(launched, ent.get("kernel_full", "")) else None   # PMC numbers of another kernel are not reported
        rows[k] = {"kernel": name, "launched": launched, "avg_us": round(avg_s * 1e6, 1), "launches": int(cnt[k]), "bound": "mfma", "achieved": round(tf, 2),
                   "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "time_share": round(ms[k] / 1e3 / (dt * prof_steps / a.steps), 4), "flops_per_launch": flops,
                   "traffic": tr, "hbm_gbps": None if tr is None else round(tr / avg_s / 1e9, 1),
                   "hbm_frac": None if tr is None else round(tr / avg_s / 1e9 / HBM_PEAK_GBPS, 4)}
        if k in CONV1_EXACT_BYTES and "_exact_kernel" in (launched or ""):
            # conv1 as exact uint8 x three-term-bf16 products on the bf16 matrix cores: 3/16 of the fp32 kernel's matrix time, so the layer's BYTES bound it
            gbps = CONV1_EXACT_BYTES[k] / avg_s / 1e9
            rows[k].update({"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                            "bytes_per_launch": CONV1_EXACT_BYTES[k], "executed_bf16_mfma_tflops": round(3 * tf, 1),
                            "executed_frac_of_bf16_mfma_peak": round(3 * tf / BF16_MFMA_PEAK_TFLOPS, 4),
                            "algorithmic_tflops": round(tf, 2)})
    if not rows:
        return None
    dom = max(rows, key=lambda k: rows[k]["time_share"])
    r = rows[dom]
    whole_tf = ALG_FLOPS_PER_ENV_STEP * (a.steps * T * E) / dt / 1e12
    exec_tf = EXEC_FLOPS_PER_ENV_STEP * (a.steps * T * E) / dt / 1e12
    out = {"bound": r["bound"], "kernel": r["kernel"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
           "traffic": r["traffic"], "hbm_gbps": r["hbm_gbps"], "hbm_frac": r["hbm_frac"], "launches": r["launches"], "avg_us": r["avg_us"],
           "flops_per_launch": r["flops_per_launch"], "time_share": r["time_share"],
           "selection": "largest measured time share; source = HIP events on the learner stream around every launch of kernel ids 0-11 during the "
                        "first %d of the %d timed steps.  Without a concurrent rollout the two clocks agree within 1 %% on all twelve (profiles/r04_microbench.txt: "
                        "events, profiles/r04_learner_only_kernel_stats.md: rocprofv3 --kernel-trace of the same minibatches); under the rollout rocprofv3 "
                        "(profiles/r04_bench_kernel_stats.md) agrees within ~3 %% for ten and reads conv1_wgrad / dense_fwd / conv3_wgrad — the kernels that leave LDS "
                        "free for actor blocks — 10-30 %% longer, because under the profiler the two queues interleave differently (actor kernels stretch to 4x "
                        "their isolated time).  The unprofiled event times are the ones that add up to the measured step, so they are what this object reports" % (prof_steps, a.steps),
           "event_steps": prof_steps,
           "min_frac": min(v["frac"] for k, v in rows.items() if k != 4),   # each kernel against ITS bound (id 4, the 32-wide heads wgrad, is 0.2 % of the flops)
           "min_frac_mfma_bound": min([v["frac"] for k, v in rows.items() if k != 4 and v["bound"] == "mfma"] or [None]),
           "whole_step": {"executed_flops_per_env_step": round(EXEC_FLOPS_PER_ENV_STEP), "executed_achieved": round(exec_tf, 2),
                          "executed_frac": round(exec_tf / FP32_MFMA_PEAK_TFLOPS, 4),
                          "algorithmic_flops_per_env_step": ALG_FLOPS_PER_ENV_STEP, "achieved": round(whole_tf, 2), "frac": round(whole_tf / FP32_MFMA_PEAK_TFLOPS, 4),
                          "algorithmic_hbm_gbps": round(ALG_BYTES_PER_ENV_STEP * (a.steps * T * E) / dt / 1e9, 1),
                          "conv1_note": "executed_* / achieved count conv1's forward and weight gradient at their ALGORITHMIC flops (2 x 25.2 GFLOP per minibatch); with "
                                        "conv1_fp32_chain = 0 those two run as three exact bf16-MFMA products each (kernels[] rows with bound = hbm), so the fraction of the "
                                        "fp32 MFMA peak is a throughput yardstick across rounds, not a utilisation of the fp32 pipe",
                          "note": "executed_*: the flops of the launched kernels (rollout forward + 4 epochs x the twelve GEMMs; no conv1 input gradient exists) "
                                  "x env-steps / wall time — the yardstick to compare rounds on; achieved / frac: SURVEY 8d's 243.2 MFLOP per env-step "
                                  "(backward priced as 2 x forward, i.e. including that non-existent conv1 dgrad) and 345.8 KB per env-step"},
           "kernels": [rows[k] for k in sorted(rows)]}
    return out


def _same_kernel(launched, profiled_full):
    """launched = '<symbol> <functor type>' from the library; profiled_full = the demangled kernel name rocprofv3 recorded.  Same kernel iff the
    symbol and the functor type both occur in the profiled name (white space differs between demanglers)."""
    if not launched or not profiled_full:
        return False
    full = profiled_full.replace(" ", "")
    return all(part.replace(" ", "") in full for part in launched.split(" ", 1))


def host_env_value(a, params):
    """Secondary value: the SAME workload through the envpool-shaped host API — numpy frames over PCIe into cbm_actor_step_host, per-step
    action D2H like the reference (ppo:317) — with the host twin of the synthetic env, `threads` actor threads.  Not the headline."""
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train
    out = {}
    for threads in (1, 2):
        warm, n_up = 2, 8
        marks = {}

        def on_update(v, stats, e, marks=marks, warm=warm, last=warm + n_up):
            if v == warm or v == last:       # two syncs per run; in between the learner thread enqueues ahead like the product run
                e.sync()
                marks[v] = time.perf_counter()

        argv = ["--local-num-envs", str(E // threads), "--num-actor-threads", str(threads), "--num-steps", str(T), "--env-backend", "host",
                "--network", "nature", "--total-timesteps", str((warm + n_up) * E * T), "--log-frequency", "100000", "--concurrency"]
        cwd = os.getcwd()
        os.chdir(os.environ.get("TMPDIR", "/tmp"))
        try:
            so = sys.stdout
            sys.stdout = open(os.devnull, "w")
            try:
                train(parse_args(argv, "ppo"), "ppo", on_update=on_update)
            finally:
                sys.stdout = so
        finally:
            os.chdir(cwd)
        out[f"actor_threads_{threads}"] = round(float(n_up * E * T / (marks[warm + n_up] - marks[warm])), 1)
    out["unit"] = "env-steps/s"
    out["note"] = ("envpool step API path: host synthetic env (persistent worker pool like envpool's, fresh obs array per step), 3.39 MB H2D + 480 B D2H and one stream "
                   "sync per 120-env step (ppo:317); total envs = 120 split over the actor threads; 8 updates between two device syncs")
    return out


def secondary_values():
    """Secondary workloads under the same clock: a few warm-up + 12 (PPO-ResNet) / 18 (backward-split) / 63 (IMPALA T = 128) / 300 (T = 20) timed updates each, in three blocks through the product trainer
    (cleanba_amd.trainer.train, device env, --concurrency), two device syncs per run.  Not the headline."""
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train
    # every row: `warm` warm-up updates, then THREE timed blocks of n_up / 3 updates each (a device sync at every block edge); value = the MEDIAN block,
    # the fastest / slowest block ride along — a 5 % move between two driver runs can then be read against the spread inside one (VERDICT r5 "next" 6).
    # A block edge drains the actor / learner pipeline: blocks are a quarter of a second or longer (PPO-ResNet timed in blocks of two updates read 63.4 k where
    # twelve updates between two syncs read 65.7 k, IMPALA T = 128 in blocks of seven 1.162 M against 1.217 M: tools/readme_table.py)
    rows = [("impala_bf16_configs2", "impala", ["--network", "nature", "--bf16-forward"], T, 5, 63,
             "BASELINE configs[2]: IMPALA a0-l0-d1, V-trace, Nature-CNN bf16-MFMA forward / fp32 returns, 120 envs x 128 steps"),
            ("impala_fp32_t128", "impala", ["--network", "nature"], T, 5, 63, "IMPALA a0-l0-d1 fp32, 120 envs x 128 steps"),
            ("impala_fp32_t20", "impala", ["--network", "nature"], 20, 40, 300, "IMPALA a0-l0-d1 fp32 at the script's default num_steps = 20 (40 warm + 300 timed updates of ~2.5 ms)"),
            ("ppo_nature_conv1_fp32_chain", "ppo", ["--network", "nature", "--conv1-fp32-chain", "3"], T, 2, 18,
             "configs[1] with conv1's forward and weight gradient on the fp32-MFMA fmaf-chain kernels (cbm_config.conv1_fp32_chain = 3: every learner-size kernel "
             "bit-identical to the CPU oracle; what rounds 1-5 measured as the headline)"),
            ("ppo_nature_backward_split2", "ppo", ["--network", "nature", "--backward-split", "2"], T, 2, 18,
             "EXTENSION, not the headline: configs[1] with the backward GEMMs as two-term split-bf16 products on bf16 MFMA, fp32 accumulate "
             "(cbm_config.backward_split = 2; gradients within 1.2e-6 of the fp32-MFMA path); the forward stays fp32 MFMA, bit-exact"),
            ("ppo_nature_backward_split3", "ppo", ["--network", "nature", "--backward-split", "3"], T, 2, 18,
             "EXTENSION, not the headline: configs[1] with the input-gradient GEMMs as three-term split-bf16 products on bf16 MFMA, fp32 accumulate "
             "(cbm_config.backward_split = 3; gradients within 1e-7 of the fp32-MFMA path, tests/test_gpu_parity.py); forward and weight gradients stay fp32 MFMA"),
            ("ppo_resnet", "ppo", ["--network", "impala_resnet"], T, 2, 12, "PPO a0-l0-d1 with the IMPALA-ResNet torso (the CLI's default network, ppo:149-189), 120 envs x 128 steps")]
    out = {}
    for name, algo, extra, t, warm, n_up, what in rows:
        marks = {}

        edges = [warm + k * (n_up // 3) for k in range(4)]

        def on_update(v, st, e, marks=marks, edges=edges):
            if v in edges:
                e.sync()
                marks[v] = time.perf_counter()

        # (three more updates than are timed: the run's last updates have no rollout beside them — the actor is done — and would make the third block
        # 5-8 % faster than the other two; rounds 3-5 timed them, which is part of why r05's ppo_resnet read 64.0 k where the median block reads 62 k)
        argv = ["--local-num-envs", str(E), "--num-actor-threads", "1", "--num-steps", str(t), "--env-backend", "device", "--total-timesteps",
                str((warm + n_up + 3) * E * t), "--log-frequency", "100000", "--concurrency"] + extra
        cwd = os.getcwd()
        os.chdir(os.environ.get("TMPDIR", "/tmp"))
        so = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            train(parse_args(argv, algo), algo, on_update=on_update)
            blk = sorted((marks[edges[k + 1]] - marks[edges[k]]) / (n_up // 3) for k in range(3))   # seconds per update, fastest block first
            dt = blk[1]
            out[name] = {"value": round(E * t / dt, 1), "unit": "env-steps/s", "ms_per_update": round(dt * 1e3, 3), "updates_timed": n_up,
                         "value_is": "median of three timed blocks of %d updates" % (n_up // 3),
                         "block_values_min_max": [round(E * t / blk[2], 1), round(E * t / blk[0], 1)], "workload": what}
        except BaseException as e:  # noqa: BLE001
            out[name] = {"value": None, "error": f"{type(e).__name__}: {e}", "workload": what}
        finally:
            sys.stdout = so
            os.chdir(cwd)
    r = out.get("ppo_resnet", {})
    if r.get("value"):
        # executed flops per env-step of the ResNet PPO step (rollout forward + 4 epochs x (forward + input gradients except conv0's + weight gradients))
        r["executed_mflop_per_env_step"] = RESNET_EXEC_MFLOP_PER_ENV_STEP
        r["executed_frac_of_fp32_mfma_peak"] = round(r["value"] * RESNET_EXEC_MFLOP_PER_ENV_STEP * 1e6 / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        r["per_kernel"] = "profiles/r06_resnet_roofline.md (tools/resnet_roofline.py over the rocprofv3 --kernel-trace of tools/rn_microbench.py: executed flops per launch from each kernel's geometry)"
    return out


def _resnet_exec_mflop():
    """Executed MFLOP per env-step of PPO on the IMPALA-ResNet torso (15 convs 3x3 SAME, dense 3872 -> 256, heads): one rollout forward + 4 epochs
    x (forward + input gradient of every conv but the first + weight gradient of every conv) + dense / heads forward, dgrad, wgrad."""
    convs = []   # (H, CI, CO)
    h, ci = 84, 4
    for co in (16, 32, 32):
        convs.append((h, ci, co))
        h = (h + 1) // 2
        convs += [(h, co, co)] * 4
        ci = co
    fwd = sum(2.0 * hh * hh * 9 * a_ * b_ for hh, a_, b_ in convs)
    dgrad = sum(2.0 * hh * hh * 9 * a_ * b_ for hh, a_, b_ in convs[1:])
    dense = 2.0 * 3872 * 256 + 2.0 * 256 * (A + 1)
    per_frame_fwd = fwd + dense
    per_frame_bwd = dgrad + fwd + 2 * dense
    return round((per_frame_fwd + EPOCHS * (per_frame_fwd + per_frame_bwd)) / 1e6, 1)


RESNET_EXEC_MFLOP_PER_ENV_STEP = _resnet_exec_mflop()


def allreduce_ab(ctx, rdv, world, rank, iters=20):
    """A/B of the two gradient all-reduce backends on IDENTICAL data inside the same N rank processes: the backend the run
    used stays in communicator slot 0, the other one is brought up in the spare slot; each all-reduces an integer-valued pattern (every order of
    summation gives the same fp32 result, so 'exact' means: every element of every rank arrived exactly once — what a stale cache line or a missed
    flag would break) and is then timed over `iters` blocking all-reduces of the 6.78 MB flat gradient.  `equal` = the two results agree bit for
    bit.  RCCL needs one device per rank; the native backend across devices needs fine-grained gradient windows (CBM_NATIVE_FINEGRAINED=1 or
    CBM_COMM=native when the contexts are created) — an unavailable backend is reported as such, never guessed at."""
    from cleanba_amd import topology
    P = ctx.P
    idx = np.arange(P, dtype=np.int64)

    def pattern(r):
        return (((idx * 2654435761 + r * 40503) >> 7) % 1024 - 512).astype(np.float32)

    want = np.zeros(P, np.float64)
    for r in range(world):
        want += pattern(r)
    want = want.astype(np.float32)
    primary = ctx.comm_backend()
    one_device = os.environ.get("CBM_FORCE_DEVICE") is not None
    out, results = {"bytes": int(P * 4), "ranks": world, "iters": iters, "primary": primary}, {}
    saved_timeout = os.environ.get("CBM_NATIVE_TIMEOUT_S")
    os.environ["CBM_NATIVE_TIMEOUT_S"] = os.environ.get("CBM_AB_TIMEOUT_S", "20")   # (read at every native launch: a stuck A/B costs seconds, not the default 120)
    try:
        _allreduce_ab_backends(ctx, rdv, world, rank, iters, P, pattern, want, primary, one_device, out, results)
    finally:   # (the topology phase that follows in the same process gets its own dead-peer timeout back)
        if saved_timeout is None:
            os.environ.pop("CBM_NATIVE_TIMEOUT_S", None)
        else:
            os.environ["CBM_NATIVE_TIMEOUT_S"] = saved_timeout
    if "rccl" in results and "native" in results:
        eq = bool(np.array_equal(results["rccl"], results["native"]))
        out["equal"] = bool(ctx.comm_allreduce_f64([1.0 if eq else 0.0], "min")[0] == 1.0)
    out["note"] = ("blocking all-reduce of the whole flat gradient, host-timed (max over ranks), nothing else on the GPU; integer-valued test pattern: 'exact' = equals "
                   "the analytic sum on every rank, 'equal' = the two backends agree bit for bit; backward_*: one learner minibatch's backward pass on the learner "
                   "stream alone and beside ONE whole-gradient all-reduce on the communication stream (HIP events, max over ranks) — what hiding the all-reduce costs")
    return out


def _allreduce_ab_backends(ctx, rdv, world, rank, iters, P, pattern, want, primary, one_device, out, results):
    for be in ("rccl", "native"):
        which = L.COMM_LEARNERS if be == primary else L.COMM_WORLD
        if be != primary:
            if be == "rccl" and one_device:
                out[be] = {"available": False, "why": "RCCL takes one rank per device; the ranks of this run share GPU %s (CBM_FORCE_DEVICE)" % os.environ["CBM_FORCE_DEVICE"]}
                continue
            if be == "native" and not one_device and os.environ.get("CBM_NATIVE_FINEGRAINED") != "1" and os.environ.get("CBM_COMM") != "native":
                out[be] = {"available": False, "why": "across devices the native kernels need fine-grained gradient windows: run with CBM_NATIVE_FINEGRAINED=1"}
                continue
            if primary == "loopback":
                out[be] = {"available": False, "why": "CBM_COMM_LOOPBACK run"}
                continue
            ranks = list(range(world))
            err = ""
            try:
                if be == "native":
                    blob = ctx.comm_native_export(which)
                    rdv.put(f"ab/native/{rank}", blob)
                    ctx.comm_native_init([blob if i == rank else bytes(rdv.get(f"ab/native/{i}")) for i in ranks], rank, which)
                else:
                    uid = rdv.share("ab/uid", ctx.comm_unique_id, 0)
                    L.Context.comm_init(ctx, which, uid, world, rank)
            except Exception as e:  # noqa: BLE001
                err = f"rank {rank}: {type(e).__name__}: {e}"
            # every rank learns whether EVERY rank brought the second backend up before anybody enters a collective on it (a rank that could not map
            # a peer must not leave the others waiting inside a kernel)
            rdv.put(f"ab/{be}/up/{rank}", err.encode() or b"ok")
            ups = [bytes(rdv.get(f"ab/{be}/up/{i}")).decode() for i in ranks]
            bad = [u for u in ups if u != "ok"]
            if bad:
                out[be] = {"available": False, "why": "could not be brought up on every rank: " + bad[0][:400]}
                continue
        ctx.write("grads", pattern(rank))
        ctx.sync()
        ctx.comm_allreduce_grads(which)
        got = ctx.read("grads", np.float32)
        results[be] = got
        ctx.comm_barrier(which)
        t0 = time.perf_counter()
        for _ in range(iters):
            ctx.comm_allreduce_grads(which)
        dt = time.perf_counter() - t0
        us = float(ctx.comm_allreduce_f64([dt / iters * 1e6], "max", which)[0])
        exact = bool(np.array_equal(got, want))
        exact_all = bool(ctx.comm_allreduce_f64([1.0 if exact else 0.0], "min", which)[0] == 1.0)
        out[be] = {"available": True, "us_per_allreduce": round(us, 1), "busbw_gbps": round(2.0 * (world - 1) / world * P * 4 / (us * 1e-6) / 1e9, 1),
                   "exact_on_every_rank": exact_all}
        # the learner-stream cost of hiding this backend's all-reduce under a backward pass (VERDICT r5 "next" 3a)
        try:
            alone, beside, ar_us = ctx.comm_overlap_probe(which, 8)
            alone, beside, ar_us = (float(ctx.comm_allreduce_f64([v], "max", which)[0]) for v in (alone, beside, ar_us))
            out[be].update({"backward_ms_alone": round(alone, 3), "backward_ms_beside_allreduce": round(beside, 3),
                            "learner_stream_slowdown_pct": round(100.0 * (beside - alone) / alone, 1), "us_per_allreduce_beside_backward": round(ar_us, 1)})
        except Exception as e:  # noqa: BLE001
            out[be]["overlap_probe_error"] = f"{type(e).__name__}: {e}"[:300]


def baseline_config_phase(a, world, rank):
    """`bench.py --gpus 4` / `--gpus 8` (the driver's SCALE command) measures the weak-scaling a0_l0_dN line; BASELINE.json's multi-GPU configs
    are split topologies — configs[3] `a0-l1,2,3` on 4 GPUs, configs[4] `2x(a0-l1,2,3)` + Atari-57 mix on 8.  With exactly 4 / 8 ranks the SAME
    processes run that topology afterwards (own rendezvous prefix) and its line rides along as `baseline_config`.  Guarded by a watchdog: the
    headline line is never lost to a failure here."""
    import copy
    import threading
    from cleanba_amd import topology
    a2 = copy.copy(a)
    a2.topology, a2.env_id = ("a0-l1,2,3", "Breakout-v5") if world == 4 else ("2x(a0-l1,2,3)", "Atari57Mix-v5")
    a2.steps, a2.warmup, a2.actor_threads = min(a.steps, 8), 2, 1
    os.environ["CBM_RDV_PREFIX"] = "cbm-topo"
    rdv = topology.Rendezvous(world, rank, os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "29500"), prefix="cbm-topo-line")
    box = {}

    def work():
        try:
            box["line"] = run_topology(a2, world, rank)
        except BaseException as e:  # noqa: BLE001
            box["error"] = f"{type(e).__name__}: {e}"

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(300.0)
    if th.is_alive():
        box["error"] = "timeout after 300 s"
    if box.get("line") is not None and rank != 0:
        rdv.put("line", json.dumps(box["line"]).encode())
    elif box.get("error") and rank != 0:
        try:
            rdv.put(f"error/{rank}", box["error"].encode())
        except Exception:  # noqa: BLE001
            pass
    if rank != 0:
        return None
    if box.get("error"):
        return {"config": a2.topology, "value": None, "error": box["error"]}
    t0 = time.time()
    while time.time() - t0 < 30.0:
        if rdv.store.check(["line"]):
            d = json.loads(bytes(rdv.store.get("line")).decode())
            return {"config": f"BASELINE configs[{3 if world == 4 else 4}] PPO {a2.topology}" + (" , Atari-57 synthetic frame mix" if world == 8 else ""),
                    "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "n_gpus": d["n_gpus"],
                    "scaling": d["scaling"], "allreduce": d.get("allreduce"), "workload": d["config"]["workload"]}
        time.sleep(0.05)
    return {"config": a2.topology, "value": None, "error": "no line from learner 0 within 30 s"}


# --------------------------------------------------------------------------------------------- split-topology bench
def parse_topology(spec):
    """'a0-l1,2,3' -> (groups=1, [0], [1,2,3]);  '2x(a0-l1,2,3)' -> (2, [0], [1,2,3]);  'a0,1-l2,3' -> (1, [0,1], [2,3])."""
    m = re.fullmatch(r"(?:(\d+)x\()?a([\d,]+)-l([\d,]+)\)?", spec.replace(" ", ""))
    if not m:
        raise SystemExit(f"--topology {spec!r}: expected dp | a<ids>-l<ids> | <G>x(a<ids>-l<ids>), e.g. a0-l1,2,3 or 2x(a0-l1,2,3)")
    return int(m.group(1) or 1), [int(x) for x in m.group(2).split(",")], [int(x) for x in m.group(3).split(",")]


def run_topology(a, world, rank):
    """BASELINE configs[3]/[4] through the product trainer (cleanba_amd.trainer.train): one process per role, the actor writes shards into
    the learners' rings (HIP IPC peer writes), gradients are all-reduced over every learner of every group (RCCL).  Timed on learner 0 of
    group 0 between the completions of update `warmup` and update `warmup + steps` (device-synchronised), max over the learner ranks."""
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train
    from cleanba_amd import topology
    groups, aids, lids = parse_topology(a.topology)
    G = len(aids) + len(lids)
    if world != groups * G:
        raise SystemExit(f"--topology {a.topology} needs {groups * G} role processes, got world size {world}")
    argv = ["--local-num-envs", str(E), "--num-actor-threads", str(a.actor_threads), "--num-steps", str(T), "--env-backend", "host" if a.dry_run else "device", "--network", "nature",
            "--env-id", a.env_id, "--total-timesteps", str((a.warmup + a.steps) * E * T * a.actor_threads * len(aids) * groups),
            "--log-frequency", "100000", "--concurrency", "--distributed", "--actor-device-ids"] + [str(i) for i in aids] + \
           ["--learner-device-ids"] + [str(i) for i in lids]
    args = parse_args(argv, "ppo")
    lay = topology.Layout(args, world, rank)
    if groups > 1 and os.environ.get("CBM_FORCE_DEVICE") is None and "CBM_GROUP_DEVICE_STRIDE" not in os.environ:
        os.environ["CBM_GROUP_DEVICE_STRIDE"] = str(max(aids + lids) + 1)   # one launcher, one visible device set: group g sits on GPUs [g*stride, ...)
    os.environ["LOCAL_RANK"] = str(lay.device_id)
    marks, seen = {}, {}

    def on_update(v, stats, engine):
        if v == a.warmup or v == a.warmup + a.steps:
            engine.sync()
            marks[v] = time.perf_counter()
            seen["backend"], seen["ranks"] = engine.comm_backend(), engine.comm_size()

    cwd = os.getcwd()
    os.chdir(os.environ.get("TMPDIR", "/tmp"))
    so = sys.stdout
    sys.stdout = sys.stderr
    try:
        res = train(args, "ppo", on_update=on_update, **({"engine_factory": _dry_engine()} if a.dry_run else {}))
    finally:
        sys.stdout = so
        os.chdir(cwd)
    if lay.is_actor or lay.group != 0 or lay.learner_index != 0:
        return None
    dt = marks[a.warmup + a.steps] - marks[a.warmup]
    env_steps = a.steps * E * T * a.actor_threads * len(aids) * groups
    return {"metric": "env-steps/sec (whole node), Breakout-v5 84x84x4, num_envs=120", "value": round(env_steps / dt, 1), "unit": "env-steps/s",
            "n_gpus": len(set(aids + lids)) * groups, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if groups == 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PPO {a.topology}: Nature-CNN fp32, local_num_envs={E} x {a.actor_threads} actor thread(s) per actor GPU, rollout_len={T}, "
                                   f"4 epochs x 4 minibatches, A=18, device synthetic env ({a.env_id}), concurrency on; {groups} group(s) of "
                                   f"{len(aids)} actor + {len(lids)} learner role processes",
                       "global_batch": E * T * a.actor_threads * len(aids) * groups, "parallelism": f"{groups}x(actor{len(aids)}+dp{len(lids)})"},
            "roofline": None, "role_processes": world, "timed_on": "learner 0 of group 0 (update completions, device-synchronised)",
            "allreduce": {"backend": seen.get("backend") or "none (one learner)", "ranks": seen.get("ranks", 0),
                          "note": "native = the library's own two-shot all-reduce over IPC-mapped peer buffers (csrc/comm.hip); rccl = RCCL over xGMI"},
            "all_roles_on_gpu": os.environ.get("CBM_FORCE_DEVICE"), "updates": int(res["updates"]), **({"dry_run": True} if a.dry_run else {})}


# --------------------------------------------------------------------------------------------- launcher
def self_launch(n, argv):
    """`python bench.py --gpus N` without torchrun: start the N rank processes here (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) on this
    process's ORIGINAL stdout (exactly one of them emits the result line), wait for all, return the worst exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=_REAL_STDOUT if _REAL_STDOUT is not None else None))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--topology", default="dp", help="dp (a0_l0_dN, the default) | a0-l1,2,3 | 2x(a0-l1,2,3) | a0-l0,1 ...: one process per role")
    ap.add_argument("--env-id", default="Breakout-v5", help="Breakout-v5 | Atari57Mix-v5 (BASELINE configs[4])")
    ap.add_argument("--actor-threads", type=int, default=1, help="actor threads per actor GPU in the split topologies")
    ap.add_argument("--prof-steps", type=int, default=0, help="timed steps that carry the per-launch HIP events (0: an eighth of --steps)")
    ap.add_argument("--prof-kernel", type=int, default=-2, help="-2: HIP events around every GEMM launch (ids 0-11, default); k: only kernel k; -1: off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-env", action="store_true", help="skip the secondary envpool-API measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (IMPALA configs[2], IMPALA fp32, PPO-ResNet)")
    ap.add_argument("--no-allreduce-ab", action="store_true", help="with N > 1 ranks: skip the RCCL / native all-reduce A/B after the timed steps")
    ap.add_argument("--no-baseline-config", action="store_true", help="with 4 / 8 ranks: skip the BASELINE configs[3] / configs[4] topology line after the dp line")
    ap.add_argument("--dry-run", action="store_true", help="CPU plumbing run of the same multi-rank code on the oracle engine over gloo (tiny sizes; never a measurement)")
    ap.add_argument("--conv1-fp32-chain", type=int, default=0, choices=[0, 1, 2, 3],
                    help="cbm_config.conv1_fp32_chain: 0 = learner-size conv1 as exact products on the bf16 matrix cores (default), 3 = the fp32-MFMA chain kernels")
    ap.add_argument("--bwd-split", type=int, default=0, choices=[0, 2, 3],
                    help="build-only extension (NOT the headline): backward GEMMs on split-bf16 MFMA, see cbm_config.backward_split")
    a = ap.parse_args()
    if a.dry_run:
        global E, T
        E, T = 12, 4              # (12 envs: three learners x four minibatches in the topology phase)
        a.prof_kernel, a.no_host_env, a.no_secondary, a.no_cpu_baseline = -1, True, True, True
    if a.topology != "dp":
        groups, aids, lids = parse_topology(a.topology)
        want_world = groups * (len(aids) + len(lids))
    else:
        want_world = a.gpus
    if "WORLD_SIZE" not in os.environ and want_world > 1:
        sys.exit(self_launch(want_world, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and os.environ.get("CBM_FORCE_DEVICE") is None:
        # one rank per GPU: the gradient windows are allocated fine-grained so that the native all-reduce can run ACROSS devices in the A/B after the
        # timed steps (allreduce_ab); measured cost on one GPU: none (33.56 vs 33.56 ms pipelined, profiles/r05_finegrained_window.txt)
        os.environ.setdefault("CBM_NATIVE_FINEGRAINED", "1")
    if a.topology == "dp" and world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: running the {world} ranks that were launched", file=sys.stderr)
    try:
        if a.topology != "dp":
            line = run_topology(a, world, rank)
            if line is not None:
                _emit(json.dumps(line))
            return
        line, params = run_dp(a, world, rank, local_rank)
    except BaseException as e:  # noqa: BLE001  a failed rank (RCCL init, IPC mapping, a dead peer) still leaves ONE parsable line on stdout
        if rank == 0 and not isinstance(e, SystemExit):
            _emit(json.dumps({"error": f"{type(e).__name__}: {e}", "n_gpus": world, "topology": a.topology, "rank": rank,
                              "metric": "env-steps/sec (whole node), Breakout-v5 84x84x4, num_envs=120", "value": None}))
        raise
    if world in (4, 8) and a.topology == "dp" and not a.no_baseline_config:
        bc = baseline_config_phase(a, world, rank)
        if rank == 0:
            line["baseline_config"] = bc
    if rank == 0:
        if world == 1 and not a.no_host_env:
            line["host_env"] = host_env_value(a, params)
            # the envpool-step-API figure (the path north_star calls the drop-in) as a named second figure beside the headline
            line["config"]["envpool_api_env_steps_per_s"] = line["host_env"].get("actor_threads_1")
            line["envpool_api_env_steps_per_s"] = {"actor_threads_1": line["host_env"].get("actor_threads_1"),
                                                   "actor_threads_2": line["host_env"].get("actor_threads_2"), "unit": "env-steps/s"}
        if world == 1 and not a.no_secondary:
            line["secondary"] = secondary_values()
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        _emit(json.dumps(line))
    if world in (4, 8):
        os._exit(0)   # (daemon threads of a timed-out topology phase must not keep the rank alive)


_REAL_STDOUT = None


def _emit(text):
    """The one JSON line goes to the process's ORIGINAL stdout; see _quiet_stdout."""
    if _REAL_STDOUT is None:
        print(text, flush=True)
    else:
        os.write(_REAL_STDOUT, (text + "\n").encode())


def _quiet_stdout():
    """Everything else that writes to fd 1 — RCCL's start-up banner (C stdio, flushed at exit, i.e. AFTER the result line), library warnings,
    the other ranks under torchrun — is sent to stderr, so stdout carries exactly one line: the result JSON."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


if __name__ == "__main__":
    _quiet_stdout()
    main()
