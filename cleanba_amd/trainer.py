"""Sebulba actor-learner host loop — the Python counterpart of cleanba_ppo.py / cleanba_impala.py's
`rollout()` (ppo:226-406, impala:268-446) and `__main__` learner loop (ppo:691-751, impala:684-760),
driving the HIP library through its C ABI (cleanba_amd.lib.Context).

What stays the same as the reference: CLI flags, thread topology (one host thread per actor slot, learner on
the main thread), step order, storage fields, `global_step` accounting (ppo:311), the `SPS:` print (ppo:383),
scalar names, policy-version skew with --concurrency (ppo:287-304), per-minibatch gradient
all-reduce across learner processes (pmean, ppo:628) — here RCCL through torch.distributed.
What is different by design: rollout data never leaves HBM (ring slots instead of Queue payloads), and with
`--env-backend device` the env itself steps on the GPU so a whole rollout is enqueued without host syncs.
"""
import json
import os
import threading
import time
import uuid
from collections import deque
from types import SimpleNamespace

import numpy as np

from . import lib as L
from . import model as M
from . import prng
from .args import distributed_env, finalize
from .envs import make_env


class JsonlWriter:
    """Stand-in for tensorboardX.SummaryWriter (not installed): same add_scalar/add_text calls.  Every scalar goes to a TensorBoard
    event file (cleanba_amd.tb) AND to runs/{run_name}/scalars.jsonl (one JSON line each, greppable)."""

    def __init__(self, logdir):
        from .tb import SummaryWriter
        os.makedirs(logdir, exist_ok=True)
        self.f = open(os.path.join(logdir, "scalars.jsonl"), "a")
        self.tb = SummaryWriter(logdir)
        self.lock = threading.Lock()

    def add_scalar(self, tag, value, step):
        with self.lock:
            self.f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step), "t": time.time()}) + "\n")
        self.tb.add_scalar(tag, value, step)

    def add_text(self, tag, text):
        with self.lock:
            self.f.write(json.dumps({"tag": tag, "text": text}) + "\n")
        self.tb.add_text(tag, text)

    def close(self):
        self.f.close()
        self.tb.close()


def make_config(args, algo):
    cfg = L.default_config(L.ALGO_PPO if algo == "ppo" else L.ALGO_IMPALA)
    cfg.device = args.learner_device_ids[0] if not args.distributed else int(os.environ.get("LOCAL_RANK", os.environ.get("SLURM_LOCALID", 0)))
    cfg.network = {"nature": L.NET_NATURE, "impala_resnet": L.NET_IMPALA_RESNET}[args.network]
    cfg.num_actions = args.num_actions
    cfg.forward_bf16 = int(bool(getattr(args, "bf16_forward", False)))
    cfg.backward_split = int(getattr(args, "backward_split", 0) or 0)
    cfg.actor_dense_ksplit = 14 if args.network == "nature" else 11  # K segments of the flatten->dense when M <= 1024 rows
    cfg.local_num_envs = args.local_num_envs
    cfg.num_actor_slots = args.num_actor_threads * len(args.actor_device_ids)
    cfg.num_steps = args.num_steps
    cfg.num_minibatches = args.num_minibatches
    cfg.grad_accum_steps = args.gradient_accumulation_steps   # optax.MultiSteps(every_k_schedule) ppo:492-500
    cfg.async_batch_size = int(getattr(args, "async_batch_size", 0) or 0)   # legacy envpool async mode (naturecnn:65-66)
    cfg.update_epochs = args.update_epochs if algo == "ppo" else 1
    cfg.norm_adv = int(args.norm_adv) if algo == "ppo" else 0
    cfg.gamma, cfg.gae_lambda = args.gamma, args.gae_lambda
    cfg.clip_coef, cfg.ent_coef, cfg.vf_coef, cfg.max_grad_norm = args.clip_coef, args.ent_coef, args.vf_coef, args.max_grad_norm
    return cfg


def rollout(key, args, algo, engine, writer, device_thread_id, world_size, process_index, stop_event, errors, on_commit=None):
    """One actor slot.  Mirrors rollout() ppo:226-406 / impala:268-446."""
    try:
        _rollout(key, args, algo, engine, writer, device_thread_id, world_size, process_index, stop_event, on_commit)
    except Exception as e:  # surface thread failures in the learner loop instead of hanging it
        errors.append(e)
        stop_event.set()
        raise


def _rollout_async(key, args, engine, writer, slot, world_size, process_index, stop_event):
    """The legacy script's rollout() with envpool in async mode (naturecnn:283-446): every recv() returns `async_batch_size` of the
    `local_num_envs` envs; a rollout is num_steps*async_update such batches, stored with their env ids; nothing is carried between rollouts."""
    E, Ba = args.local_num_envs, args.async_batch_size
    async_update = E // Ba
    env_seed = args.seed + (0 if args.same_env_seed_all_ranks else process_index)  # naturecnn:290
    engine.actor_set_key(slot, key)
    envs = make_env(args.env_id, env_seed, E, backend=args.env_backend, num_actions=args.num_actions, async_batch_size=Ba)()
    global_step = 0
    start_time = time.time()
    episode_returns = np.zeros((E,), dtype=np.float32)
    returned_episode_returns = np.zeros((E,), dtype=np.float32)
    episode_lengths = np.zeros((E,), dtype=np.float32)
    returned_episode_lengths = np.zeros((E,), dtype=np.float32)
    params_queue_get_time, rollout_time, rollout_queue_put_time = deque(maxlen=10), deque(maxlen=10), deque(maxlen=10)
    actions = np.empty(Ba, np.int32)
    envs.async_reset()
    for update in range(1, args.num_updates + 2):
        if stop_event.is_set():
            return
        update_time_start = time.time()
        env_recv_time = inference_time = storage_time = env_send_time = 0.0
        t0 = time.time()
        engine.actor_begin_rollout(slot, args.concurrency)
        params_queue_get_time.append(time.time() - t0)
        rollout_time_start = time.time()
        truncations = terminations = 0
        for _ in range(async_update, (args.num_steps + 1) * async_update):   # naturecnn:343-345
            t1 = time.time()
            next_obs, next_reward, next_done, info = envs.recv()
            env_recv_time += time.time() - t1
            global_step += len(next_done) * len(args.actor_device_ids) * world_size
            env_id = info["env_id"]
            t1 = time.time()
            engine.actor_step_async(slot, next_obs, next_reward, next_done, env_id, actions)
            inference_time += time.time() - t1
            t1 = time.time()
            envs.send(actions, env_id)
            env_send_time += time.time() - t1
            t1 = time.time()
            truncated = info["elapsed_step"] >= envs.spec.config.max_episode_steps
            ended = (info["terminated"] + truncated) > 0
            truncations += int(np.sum(truncated))
            terminations += int(np.sum(info["terminated"]))
            episode_returns[env_id] += info["reward"]
            returned_episode_returns[env_id] = np.where(ended, episode_returns[env_id], returned_episode_returns[env_id])
            episode_returns[env_id] *= (1 - info["terminated"]) * (1 - truncated)
            episode_lengths[env_id] += 1
            returned_episode_lengths[env_id] = np.where(ended, episode_lengths[env_id], returned_episode_lengths[env_id])
            episode_lengths[env_id] *= (1 - info["terminated"]) * (1 - truncated)
            storage_time += time.time() - t1
        rollout_time.append(time.time() - rollout_time_start)
        t0 = time.time()
        engine.actor_commit(slot, None, None)
        rollout_queue_put_time.append(time.time() - t0)
        if update % args.log_frequency == 0:
            avg_episodic_return = float(np.mean(returned_episode_returns))
            print(f"global_step={global_step}, avg_episodic_return={avg_episodic_return}")
            print("SPS:", int(global_step / (time.time() - start_time)))
            for tag, val in (("stats/rollout_time", np.mean(rollout_time)), ("charts/avg_episodic_return", avg_episodic_return),
                             ("charts/avg_episodic_length", float(np.mean(returned_episode_lengths))),
                             ("stats/params_queue_get_time", np.mean(params_queue_get_time)), ("stats/truncations", truncations),
                             ("stats/terminations", terminations), ("stats/env_recv_time", env_recv_time),
                             ("stats/inference_time", inference_time), ("stats/storage_time", storage_time),
                             ("stats/env_send_time", env_send_time), ("stats/rollout_queue_put_time", np.mean(rollout_queue_put_time)),
                             ("charts/SPS", int(global_step / (time.time() - start_time))),
                             ("charts/SPS_update", int(E * args.num_steps * len(args.actor_device_ids) * world_size / (time.time() - update_time_start)))):
                writer.add_scalar(tag, val, global_step)


def _rollout(key, args, algo, engine, writer, slot, world_size, process_index, stop_event, on_commit=None):
    if getattr(args, "async_batch_size", 0):
        return _rollout_async(key, args, engine, writer, slot, world_size, process_index, stop_event)
    len_actor_device_ids = len(args.actor_device_ids)
    E = args.local_num_envs
    env_seed = args.seed + (0 if args.same_env_seed_all_ranks else process_index) + slot  # ppo:238
    device_env = args.env_backend == "device"
    engine.actor_set_key(slot, key)
    if device_env:
        from .envs import is_atari57_mix
        engine.actor_env_reset_device(slot, env_seed, is_atari57_mix(args.env_id))
    else:
        envs = make_env(args.env_id, env_seed, E, backend=args.env_backend, num_actions=args.num_actions)()
    global_step = 0
    start_time = time.time()
    episode_returns = np.zeros((E,), dtype=np.float32)
    returned_episode_returns = np.zeros((E,), dtype=np.float32)
    episode_lengths = np.zeros((E,), dtype=np.float32)
    returned_episode_lengths = np.zeros((E,), dtype=np.float32)
    params_queue_get_time = deque(maxlen=10)
    rollout_time = deque(maxlen=10)
    rollout_queue_put_time = deque(maxlen=10)
    actions = np.empty(E, np.int32)
    first_rollout = True
    if not device_env:
        if algo == "ppo":
            next_obs = envs.reset()
            next_done = np.zeros(E, dtype=bool)
        else:
            envs.async_reset()

    for update in range(1, args.num_updates + 2):
        if stop_event.is_set():
            return
        update_time_start = time.time()
        env_recv_time = inference_time = storage_time = d2h_time = env_send_time = 0.0
        t0 = time.time()
        actor_policy_version = engine.actor_begin_rollout(slot, args.concurrency)  # params_queue.get() + ring slot
        params_queue_get_time.append(time.time() - t0)
        rollout_time_start = time.time()
        if algo == "ppo":
            nsteps = args.num_steps
        else:
            nsteps = args.num_steps + 1 if first_rollout else args.num_steps  # impala:327-329
        step_inc = E * args.num_actor_threads * len_actor_device_ids * world_size
        if device_env:
            t1 = time.time()
            engine.actor_rollout_device(slot, nsteps)
            inference_time += time.time() - t1
            global_step += nsteps * step_inc
        else:
            for _ in range(nsteps):
                global_step += step_inc
                if algo == "ppo":
                    t1 = time.time()
                    engine.actor_step_host(slot, next_obs, next_done, None, None, actions)
                    inference_time += time.time() - t1
                    t1 = time.time()
                    next_obs, next_reward, next_done, info = envs.step(actions)
                    env_send_time += time.time() - t1
                    t1 = time.time()
                    engine.actor_record_host(slot, next_reward)
                else:
                    t1 = time.time()
                    next_obs, next_reward, next_done, info = envs.recv()
                    env_recv_time += time.time() - t1
                    t1 = time.time()
                    engine.actor_step_host(slot, next_obs, next_done, info["elapsed_step"] == 0, next_reward, actions)
                    inference_time += time.time() - t1
                    t1 = time.time()
                    envs.send(actions, info["env_id"])
                    env_send_time += time.time() - t1
                    t1 = time.time()
                env_id = info["env_id"]
                truncated = info["elapsed_step"] >= envs.spec.config.max_episode_steps  # ppo:328
                ended = (info["terminated"] + truncated) > 0
                episode_returns[env_id] += info["reward"]
                returned_episode_returns[env_id] = np.where(ended, episode_returns[env_id], returned_episode_returns[env_id])
                episode_returns[env_id] *= (1 - info["terminated"]) * (1 - truncated)
                episode_lengths[env_id] += 1
                returned_episode_lengths[env_id] = np.where(ended, episode_lengths[env_id], returned_episode_lengths[env_id])
                episode_lengths[env_id] *= (1 - info["terminated"]) * (1 - truncated)
                storage_time += time.time() - t1
        rollout_time.append(time.time() - rollout_time_start)
        first_rollout = False

        t0 = time.time()
        if device_env or algo != "ppo":
            engine.actor_commit(slot, None, None)
        else:
            engine.actor_commit(slot, next_obs, next_done)  # next_obs / next_done are still on the host (ppo:361-363)
        rollout_queue_put_time.append(time.time() - t0)
        if on_commit is not None:  # split topology: ship this slot's shards to the learner processes
            on_commit(slot, update, engine.actor_ring_index(slot))

        if update % args.log_frequency == 0:
            if device_env:
                avg_episodic_return, avg_len = engine.actor_episode_stats(slot)
            else:
                avg_episodic_return, avg_len = float(np.mean(returned_episode_returns)), float(np.mean(returned_episode_lengths))
            if slot == 0:
                print(f"global_step={global_step}, avg_episodic_return={avg_episodic_return}, rollout_time={np.mean(rollout_time)}")
                print("SPS:", int(global_step / (time.time() - start_time)))
            writer.add_scalar("stats/rollout_time", np.mean(rollout_time), global_step)
            writer.add_scalar("charts/avg_episodic_return", avg_episodic_return, global_step)
            writer.add_scalar("charts/avg_episodic_length", avg_len, global_step)
            writer.add_scalar("stats/params_queue_get_time", np.mean(params_queue_get_time), global_step)
            writer.add_scalar("stats/env_recv_time", env_recv_time, global_step)
            writer.add_scalar("stats/inference_time", inference_time, global_step)
            writer.add_scalar("stats/storage_time", storage_time, global_step)
            writer.add_scalar("stats/d2h_time", d2h_time, global_step)
            writer.add_scalar("stats/env_send_time", env_send_time, global_step)
            writer.add_scalar("stats/rollout_queue_put_time", np.mean(rollout_queue_put_time), global_step)
            writer.add_scalar("charts/SPS", int(global_step / (time.time() - start_time)), global_step)
            writer.add_scalar("charts/SPS_update", int(E * args.num_steps * len_actor_device_ids * args.num_actor_threads * world_size /
                                                       (time.time() - update_time_start)), global_step)


class GradAllReducer:
    """pmean(grads) over all learner processes (ppo:628) = all-reduce(SUM) of the library's flat gradient buffer, divided by the world
    size inside the optimizer kernel.  RCCL via torch.distributed on the HIP engine, gloo on CPU in tests.

    On the HIP engine the all-reduce is split in two so that most of it hides under the backward pass: the gradient is produced from the
    back, and the tail [grad_tail_offset, P) — dense layer + heads, 95 % of the bytes — is final long before the conv kernels are done.
    `cbm_learner_minibatch_grad` only enqueues work, so by the time this is called the GPU is still inside the backward pass: the tail is
    all-reduced on a communication stream that waits for the library's tail event, the small head on the learner stream after the
    backward pass, and the learner stream joins the communication stream before the optimizer step.  `overlap=False` (or an engine without
    the hooks) gives the single flat all-reduce."""

    def __init__(self, engine, world_size, group=None, dist_module=None, overlap=None, active=None):
        self.engine, self.world, self.group = engine, world_size, group
        self.active = world_size > 1 if active is None else bool(active)   # active=True at world 1: exercise the path on one GPU
        self.tensor = engine.grads_tensor() if self.active else None
        self.dist = dist_module
        want = os.environ.get("CBM_ALLREDUCE_OVERLAP", "1") != "0" if overlap is None else overlap
        self.overlap = bool(want and self.active and hasattr(engine, "learner_stream_wait_tail"))
        if self.overlap:
            import torch
            self.comm = torch.cuda.Stream(device=self.tensor.device)
            self.tail = engine.grad_tail_offset()

    def __call__(self):
        if self.active:
            dist = self.dist
            if dist is None:
                import torch.distributed as dist
            if self.overlap:
                import torch
                with torch.cuda.stream(self.comm):
                    self.engine.learner_stream_wait_tail(self.comm.cuda_stream)
                    dist.all_reduce(self.tensor[self.tail:], op=dist.ReduceOp.SUM, group=self.group)
                with self.engine.stream_context():
                    dist.all_reduce(self.tensor[:self.tail], op=dist.ReduceOp.SUM, group=self.group)
                self.engine.learner_wait_stream(self.comm.cuda_stream)
            else:
                with self.engine.stream_context():
                    dist.all_reduce(self.tensor, op=dist.ReduceOp.SUM, group=self.group)
        return float(self.world)  # grad_div: the mean is taken inside the optimizer kernel


def schedules(args, algo, opt_count, n_steps):
    """lr / bias corrections for optimizer steps opt_count .. opt_count+n_steps-1 (inject_hyperparams count)."""
    spu = args.num_minibatches * (args.update_epochs if algo == "ppo" else 1)
    lrs, b1s, b2s = [], [], []
    for i in range(n_steps):
        c = opt_count + i
        lrs.append(M.linear_schedule(c, args.learning_rate, spu, args.num_updates, args.anneal_lr))
        bc1, bc2 = M.adam_bias_corrections(c + 1)
        b1s.append(bc1)
        b2s.append(bc2)
    return np.array(lrs, np.float32), np.array(b1s, np.float32), np.array(b2s, np.float32)


def train(args, algo="ppo", engine_factory=None, on_update=None, rendezvous=None, dist_module=None):
    """The `__main__` block of cleanba_ppo.py / cleanba_impala.py (ppo:409-771).  `rendezvous` = (world, rank, local_rank, addr, port)
    and `dist_module` override the process environment / torch.distributed (used by the single-GPU loopback test of the split path)."""
    world_size, rank, local_rank, master_addr, master_port = rendezvous or (distributed_env() if args.distributed else (1, 0, 0, None, None))
    from . import topology
    lay = None
    if topology.is_split(args):  # actor GPU(s) and learner GPUs are different processes (README.md:62, benchmark.sh:80)
        if not args.distributed or world_size < 2:
            raise SystemExit("split topologies (--actor-device-ids != --learner-device-ids) run one process per GPU: launch with "
                             "torchrun / the SLURM variables and pass --distributed")
        lay = topology.Layout(args, world_size, rank)
        n_proc, proc_index = lay.groups, lay.group   # the reference's world_size counts actor+learner groups (ppo:425-430)
    else:
        n_proc, proc_index = world_size, rank
    if getattr(args, "async_batch_size", 0) and algo != "ppo":
        raise SystemExit("--async-batch-size belongs to the PPO script (cleanba_impala.py already drives envpool through recv/send)")
    finalize(args, n_proc, proc_index)
    if args.distributed and world_size > 1 and dist_module is None:
        import torch.distributed as dist
        if not dist.is_initialized():
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, init_method=f"tcp://{master_addr}:{master_port}", rank=rank, world_size=world_size)
    run_name = f"{args.env_id}__{args.exp_name}__{args.seed}__{uuid.uuid4()}"
    if args.track and rank == 0:   # ppo:447-458 — same wandb.init call when wandb is importable; never a silent no-op
        try:
            import wandb
            wandb.init(project=args.wandb_project_name, entity=args.wandb_entity, sync_tensorboard=True, config=vars(args), name=run_name,
                       monitor_gym=True, save_code=True)
        except ImportError:
            print("--track: wandb is not installed here; scalars are written to TensorBoard event files under runs/ only")
    writer = JsonlWriter(f"runs/{run_name}") if rank == 0 else SimpleNamespace(add_scalar=lambda *a: None, add_text=lambda *a: None, close=lambda: None)
    writer.add_text("hyperparameters", "|param|value|\n|-|-|\n%s" % ("\n".join([f"|{k}|{v}|" for k, v in vars(args).items()])))

    # seeding (ppo:465-470): identical model / learner keys in every process, env seeds differ by process index
    key = prng.prng_key(args.seed)
    key, network_key, actor_key, critic_key = prng.split(key, 4)
    learner_key = key.copy()

    cfg = make_config(args, algo)
    if lay is not None and not lay.is_actor:   # learner-only context: its slots are ingest ports for E/L env columns each
        cfg.local_num_envs = args.local_num_envs // lay.nl
    engine = engine_factory(cfg) if engine_factory else HipEngine(cfg)
    params = M.init_params(args.network, args.num_actions, network_key, actor_key, critic_key)
    engine.set_params(params)
    if lay is not None:
        return _train_split(args, algo, engine, lay, writer, key, rank, run_name, dist_module)
    allreduce = GradAllReducer(engine, world_size)

    dummy_writer = SimpleNamespace(add_scalar=lambda x, y, z: None)
    stop_event, errors, threads = threading.Event(), [], []
    n_slots = args.num_actor_threads * len(args.actor_device_ids)
    for slot in range(n_slots):
        th = threading.Thread(target=rollout, args=(key.copy(), args, algo, engine, writer if slot == 0 else dummy_writer, slot, world_size,
                                                    rank, stop_event, errors), daemon=True)
        th.start()
        threads.append(th)

    rollout_queue_get_time = deque(maxlen=10)
    learner_policy_version = 0
    opt_count = 0
    n_opt = args.num_minibatches * (args.update_epochs if algo == "ppo" else 1)
    epochs = args.update_epochs if algo == "ppo" else 1
    start = time.time()
    last_stats = None
    while True:
        learner_policy_version += 1
        t0 = time.time()
        engine.learner_wait()  # every actor slot's rollout (ppo:697-711)
        if errors:
            raise errors[0]
        rollout_queue_get_time.append(time.time() - t0)
        training_time_start = time.time()
        lrs, bc1, bc2 = schedules(args, algo, opt_count, n_opt)
        want_stats = learner_policy_version % args.log_frequency == 0 or learner_policy_version >= args.num_updates
        if world_size == 1:
            learner_key, stats = engine.learner_update(learner_key, lrs, bc1, bc2, want_stats)
        else:
            learner_key = engine.learner_prepare(learner_key)
            i = 0
            for e in range(epochs):
                learner_key = engine.learner_epoch_begin(learner_key)
                k = args.gradient_accumulation_steps
                for mb in range(args.num_minibatches * k):
                    engine.learner_minibatch_grad(e, mb)
                    grad_div = allreduce()
                    if k > 1:   # optax.MultiSteps: running mean of the (already pmean-ed) micro-batch gradients, step on every k-th
                        engine.learner_accumulate(mb % k, grad_div)
                        if mb % k != k - 1:
                            continue
                        grad_div = 1.0
                    engine.learner_optimizer_step(float(lrs[i]), float(bc1[i]), float(bc2[i]), grad_div)
                    i += 1
            stats = engine.learner_finish(n_opt, want_stats)
        opt_count += n_opt
        if stats is not None:
            last_stats = stats
        global_step = learner_policy_version * args.local_batch_size * world_size
        if on_update:
            on_update(learner_policy_version, stats)
        if learner_policy_version % args.log_frequency == 0 and stats is not None:
            writer.add_scalar("stats/rollout_queue_get_time", np.mean(rollout_queue_get_time), global_step)
            writer.add_scalar("stats/training_time", time.time() - training_time_start, global_step)
            print(global_step, f"learner_policy_version={learner_policy_version}, training time: {time.time() - training_time_start}s")
            writer.add_scalar("charts/learning_rate", float(lrs[-1]), global_step)
            if algo == "ppo":
                writer.add_scalar("losses/value_loss", float(stats[-1, 2]), global_step)
                writer.add_scalar("losses/policy_loss", float(stats[-1, 1]), global_step)
                writer.add_scalar("losses/entropy", float(stats[-1, 3]), global_step)
                writer.add_scalar("losses/approx_kl", float(stats[-1, 4]), global_step)
                writer.add_scalar("losses/loss", float(stats[-1, 0]), global_step)
            else:
                writer.add_scalar("losses/value_loss", float(stats[-1, 2]), global_step)
                writer.add_scalar("losses/policy_loss", float(stats[-1, 1]), global_step)
                writer.add_scalar("losses/entropy", float(stats[-1, 3]), global_step)
                writer.add_scalar("losses/loss", float(stats[-1, 0]), global_step)
        if learner_policy_version >= args.num_updates:
            break
    engine.sync()
    elapsed = time.time() - start
    stop_event.set()
    for th in threads:
        th.join(timeout=30)
    result = {"updates": learner_policy_version, "global_step": learner_policy_version * args.local_batch_size * world_size,
              "elapsed_s": elapsed, "stats": last_stats, "params": engine.get_params(), "run_name": run_name}
    if args.save_model and rank == 0:
        from .checkpoint import save_cleanrl_model
        path = f"runs/{run_name}/{args.exp_name}.cleanrl_model"
        save_cleanrl_model(path, args, result["params"], args.num_actions, args.network)
        print(f"model saved to {path}")
        result["model_path"] = path
        if engine_factory is None and args.eval_episodes > 0:   # ppo:773-785: 10 evaluation episodes logged as eval/episodic_return
            from .envs import make_env as _mk
            from .evals import evaluate
            engine.close()
            thunk = lambda env_id, seed, num_envs: _mk(env_id, seed, num_envs, backend="envpool" if args.env_backend == "envpool" else "host",
                                                       num_actions=args.num_actions)
            rets = evaluate(path, thunk, args.env_id, eval_episodes=args.eval_episodes, run_name=f"{run_name}-eval", network=args.network,
                            max_episode_steps=args.eval_max_episode_steps or None)
            for idx, r in enumerate(rets):
                writer.add_scalar("eval/episodic_return", r, idx)
            result["eval_returns"] = rets
            if args.upload_model:   # ppo:785-799
                try:
                    from cleanrl_utils.huggingface import push_to_hub
                    repo_name = f"{args.env_id}-{args.exp_name}-seed{args.seed}"
                    push_to_hub(args, rets, f"{args.hf_entity}/{repo_name}" if args.hf_entity else repo_name, "PPO" if algo == "ppo" else "IMPALA",
                                f"runs/{run_name}", f"videos/{run_name}-eval", extra_dependencies=["cleanba_amd"])
                except ImportError:
                    print("--upload-model: cleanrl_utils.huggingface is not installed here; the model stays at", path)
    writer.close()
    engine.close()
    return result


def _train_split(args, algo, engine, lay, writer, key, rank, run_name, dist_module=None):
    """One process of an actor/learner-split run (cleanba_amd.topology).  Actor ranks run the rollout threads and ship shards;
    learner ranks ingest shards, all-reduce gradients over every learner rank of every group, and learner 0 returns params."""
    if dist_module is None:
        import torch.distributed as dist
    else:
        dist = dist_module
    from . import topology
    groups = topology.Groups(dist, lay)
    n_opt = args.num_minibatches * (args.update_epochs if algo == "ppo" else 1)
    epochs = args.update_epochs if algo == "ppo" else 1
    start = time.time()
    if lay.is_actor:
        shipper = topology.ActorShipper(engine, lay, groups, args, algo, dist, args.num_updates)
        receiver = topology.ParamReceiver(engine, lay, groups, dist, args.num_updates)
        stop_event, errors, threads = threading.Event(), [], []
        dummy_writer = SimpleNamespace(add_scalar=lambda x, y, z: None)
        for slot in range(args.num_actor_threads * len(args.actor_device_ids)):
            th = threading.Thread(target=rollout, args=(key.copy(), args, algo, engine, writer if slot == 0 else dummy_writer, slot, lay.groups,
                                                        lay.group, stop_event, errors, shipper.on_commit), daemon=True)
            th.start()
            threads.append(th)
        shipper.thread.start()
        receiver.thread.start()
        shipper.thread.join()
        receiver.thread.join()
        stop_event.set()
        if errors or shipper.error or receiver.error:
            raise (errors + [shipper.error, receiver.error])[0] or RuntimeError("actor failed")
        elapsed = time.time() - start
        for th in threads:   # like the reference's actors, each slot runs one rollout past the last update before it sees the stop
            th.join(timeout=60)
        engine.sync()
        result = {"updates": args.num_updates, "elapsed_s": elapsed, "stats": None, "params": engine.get_actor_params(),
                  "run_name": run_name, "role": "actor"}
        writer.close()
        engine.close()
        return result
    allreduce = GradAllReducer(engine, len(lay.all_learner_ranks), groups.learners)
    ingest = topology.LearnerReceiver(engine, lay, groups, args, algo, dist, args.num_updates)
    ingest.thread.start()
    learner_key = key.copy()
    opt_count, stats = 0, None
    for version in range(1, args.num_updates + 1):
        engine.learner_wait()
        lrs, bc1, bc2 = schedules(args, algo, opt_count, n_opt)
        learner_key = engine.learner_prepare(learner_key)
        i = 0
        for e in range(epochs):
            learner_key = engine.learner_epoch_begin(learner_key)
            k = args.gradient_accumulation_steps
            for mb in range(args.num_minibatches * k):
                engine.learner_minibatch_grad(e, mb)
                grad_div = allreduce()
                if k > 1:
                    engine.learner_accumulate(mb % k, grad_div)
                    if mb % k != k - 1:
                        continue
                    grad_div = 1.0
                engine.learner_optimizer_step(float(lrs[i]), float(bc1[i]), float(bc2[i]), grad_div)
                i += 1
        stats = engine.learner_finish(n_opt, True)
        opt_count += n_opt
        if lay.learner_index == 0:
            with engine.stream_context():
                dist.send(engine.params_tensor(), dst=lay.actor_rank, group=groups.params[lay.group])
        if version % args.log_frequency == 0 and lay.learner_index == 0:
            print(version * args.local_batch_size * lay.groups, f"learner_policy_version={version}")
    engine.sync()
    ingest.thread.join(timeout=60)
    if ingest.error:
        raise ingest.error
    result = {"updates": args.num_updates, "elapsed_s": time.time() - start, "stats": stats, "params": engine.get_params(), "run_name": run_name,
              "role": f"learner{lay.learner_index}"}
    writer.close()
    engine.close()
    return result


class HipEngine(L.Context):
    """The product engine: cleanba_amd.lib.Context + the torch.distributed plumbing for the grad all-reduce."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self._tls = threading.local()

    def grads_tensor(self):
        import torch
        ptr, nbytes = self.buffer("grads")

        class _CAI:  # __cuda_array_interface__ view of the library-owned buffer (no copy)
            __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None}
        return torch.as_tensor(_CAI(), device=f"cuda:{self.cfg.device}")

    def stream_context(self):
        import torch
        return torch.cuda.stream(torch.cuda.ExternalStream(self.learner_stream(), device=f"cuda:{self.cfg.device}"))

    # ---- split topologies (cleanba_amd.topology): zero-copy torch views of the library-owned ring / parameter buffers
    def _view(self, name, ring, dtype, shape):
        import torch
        ptr, _ = self.buffer(name, ring)
        typestr = {"u8": "|u1", "i32": "<i4", "f32": "<f4"}[dtype]

        class _CAI:
            __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}
        return torch.as_tensor(_CAI(), device=f"cuda:{self.cfg.device}")

    def ring_tensors(self, ring):
        c = self.cfg
        T1, B, A = c.num_steps + 1, c.local_num_envs * c.num_actor_slots, c.num_actions
        return {"obs": self._view("obs", ring, "u8", (T1, B, L.FRAME)), "actions": self._view("actions", ring, "i32", (T1, B)),
                "logprobs": self._view("logprobs", ring, "f32", (T1, B)), "values": self._view("values", ring, "f32", (T1, B)),
                "rewards": self._view("rewards", ring, "f32", (T1, B)), "dones": self._view("dones", ring, "u8", (T1, B)),
                "firststeps": self._view("firststeps", ring, "u8", (T1, B)), "logits": self._view("logits", ring, "f32", (T1, B, A))}

    def actor_fence(self, slot):
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.ExternalStream(self.actor_stream(slot), device=f"cuda:{self.cfg.device}"))
        return ev

    def io_context(self):
        """A per-thread side stream for shard / parameter transfers (never the actor or learner compute streams)."""
        import torch
        if not hasattr(self._tls, "stream"):
            self._tls.stream = torch.cuda.Stream(device=f"cuda:{self.cfg.device}")
        return torch.cuda.stream(self._tls.stream)

    def io_wait(self, fence):
        self._tls.stream.wait_event(fence)

    def io_sync(self):
        self._tls.stream.synchronize()

    def params_tensor(self):
        return self._view("params", 0, "f32", (self.P,))

    def params_staging_tensor(self):
        import torch
        return torch.empty(self.P, dtype=torch.float32, device=f"cuda:{self.cfg.device}")

    def params_publish_external_tensor(self, t):
        self.params_publish_external(t.data_ptr())

    def get_actor_params(self):
        return self.read("actor_params_latest", np.float32)
