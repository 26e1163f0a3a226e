"""Sebulba actor-learner host loop — the Python counterpart of cleanba_ppo.py / cleanba_impala.py's
`rollout()` (ppo:226-406, impala:268-446) and `__main__` learner loop (ppo:691-751, impala:684-760),
driving the HIP library through its C ABI (cleanba_amd.lib.Context).

What stays the same as the reference: CLI flags, thread topology (one host thread per actor slot, learner on
the main thread), step order, storage fields, `global_step` accounting (ppo:311), the `SPS:` print (ppo:383),
scalar names, policy-version skew with --concurrency (ppo:287-304), per-minibatch gradient
all-reduce across learner processes (pmean, ppo:628) — here RCCL behind the C ABI (csrc/comm.hip), rendezvous over a TCP store.
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
    ch, hd = list(getattr(args, "channels", [16, 32, 32])), list(getattr(args, "hiddens", [256]))   # ppo:92-95; cbm_ctx_create checks them
    if len(ch) > 4 or len(hd) > 4:
        raise SystemExit("--channels/--hiddens: the HIP ResNet torso is built for channels [16, 32, 32] and one hidden layer of 64..512 units")
    cfg.num_channels, cfg.num_hiddens = len(ch), len(hd)
    for i, v in enumerate(ch):
        cfg.channels[i] = int(v)
    for i, v in enumerate(hd):
        cfg.hiddens[i] = int(v)
    cfg.forward_bf16 = int(bool(getattr(args, "bf16_forward", False)))
    cfg.backward_split = int(getattr(args, "backward_split", 0) or 0)
    cfg.conv1_fp32_chain = int(getattr(args, "conv1_fp32_chain", 0) or 0)
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


def rollout(key, args, algo, engine, writer, slot, world_size, process_index, stop_event, errors, on_commit=None, device_thread_id=None,
            on_error=None):
    """One actor thread.  Mirrors rollout() ppo:226-406 / impala:268-446.  `slot` is the thread's slot in this process's context,
    `device_thread_id` its index among ALL actor threads of the group (d_idx * num_actor_threads + thread_id, ppo:680), which seeds its envs."""
    try:
        _rollout(key, args, algo, engine, writer, slot, world_size, process_index, stop_event, on_commit,
                 slot if device_thread_id is None else device_thread_id)
    except Exception as e:  # surface thread failures in the learner loop instead of hanging it
        if not stop_event.is_set():   # (after the stop the aborted context makes the blocked begin_rollout fail on purpose)
            errors.append(e)
            stop_event.set()
            engine.abort()            # the learner blocked in cbm_learner_wait returns with an error instead of waiting forever
            if on_error is not None:  # split topologies: wake the shipper (it waits for THIS thread's commits) and tell the peers now
                on_error()
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


def _rollout(key, args, algo, engine, writer, slot, world_size, process_index, stop_event, on_commit=None, device_thread_id=0):
    if getattr(args, "async_batch_size", 0):
        return _rollout_async(key, args, engine, writer, slot, world_size, process_index, stop_event)
    len_actor_device_ids = len(args.actor_device_ids)
    E = args.local_num_envs
    env_seed = args.seed + (0 if args.same_env_seed_all_ranks else process_index) + device_thread_id  # ppo:238
    device_env = args.env_backend == "device"
    engine.actor_set_key(slot, key)
    if device_env:
        from .envs import is_atari57_mix
        engine.actor_env_reset_device(slot, env_seed, is_atari57_mix(args.env_id))
    else:
        # (reuse_buffers: this loop hands every step's results to the engine before the next step — the twin may rotate its result buffers)
        envs = make_env(args.env_id, env_seed, E, backend=args.env_backend, num_actions=args.num_actions, reuse_buffers=True)()
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
    arange_E, ident_ids = np.arange(E), [None, False]
    # An env that hands its observations back in RECURRING buffers — the same one every step (a pool that steps in place) or a few in rotation
    # (a state-buffer queue) — gets each such buffer page-locked the second time its address shows up, so the per-step 3.39 MB upload is a plain
    # DMA the host does not wait for instead of a staged pageable copy (cbm_host_register).  Arrays that never come back are left alone; at most
    # MAX_PINNED distinct buffers are registered.
    MAX_PINNED = 8
    pinned, seen = {}, {}
    register = getattr(engine, "host_register", None)

    def maybe_pin(obs):
        ptr = obs.ctypes.data
        if register is None or ptr in pinned:
            return
        n = seen.get(ptr, 0) + 1
        if n >= 2 and len(pinned) < MAX_PINNED and obs.flags.c_contiguous:
            try:
                register(obs)
                pinned[ptr] = obs          # keeps the array alive while it is registered
            except RuntimeError:
                pinned[ptr] = None         # not registrable (e.g. a read-only mapping): do not retry every step
            seen.pop(ptr, None)
            return
        if len(seen) > 64:                 # fresh arrays every step: forget the oldest addresses
            seen.pop(next(iter(seen)))
        seen[ptr] = n

    if not device_env:
        if algo == "ppo":
            next_obs = envs.reset()
            next_done = np.zeros(E, dtype=bool)
        else:
            envs.async_reset()

    try:
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
                        maybe_pin(next_obs)
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
                        maybe_pin(next_obs)
                        engine.actor_step_host(slot, next_obs, next_done, info["elapsed_step"] == 0, next_reward, actions)
                        inference_time += time.time() - t1
                        t1 = time.time()
                        envs.send(actions, info["env_id"])
                        env_send_time += time.time() - t1
                        t1 = time.time()
                    # episode bookkeeping of ppo:326-339 (same values; written with in-place numpy calls and, when the pool returns every env in id order
                    # — `env_id` is arange, checked once per array object —, without the gather / scatter through env_id)
                    env_id = info["env_id"]
                    if env_id is not ident_ids[0]:
                        ident_ids[0], ident_ids[1] = env_id, bool(env_id.shape[0] == E and np.array_equal(env_id, arange_E))
                    truncated = info["elapsed_step"] >= envs.spec.config.max_episode_steps  # ppo:328
                    ended = (info["terminated"] + truncated) > 0
                    if ident_ids[1]:
                        episode_returns += info["reward"]
                        np.copyto(returned_episode_returns, episode_returns, where=ended)
                        episode_returns[ended] = 0.0
                        episode_lengths += 1
                        np.copyto(returned_episode_lengths, episode_lengths, where=ended)
                        episode_lengths[ended] = 0.0
                    else:
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
    finally:
        # page-locked observation buffers are un-registered before their arrays can be freed: a stale registration over recycled addresses makes a
        # later upload from there fail with "invalid argument"
        unregister = getattr(engine, "host_unregister", None)
        for arr in pinned.values():
            if arr is not None and unregister is not None:
                try:
                    unregister(arr)
                except Exception:  # noqa: BLE001  (the context may already be gone on an error path)
                    pass
        pinned.clear()


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


def train(args, algo="ppo", engine_factory=None, on_update=None, rendezvous=None):
    """The `__main__` block of cleanba_ppo.py / cleanba_impala.py (ppo:409-771).  `rendezvous` = (world, rank, local_rank, addr, port)
    overrides the process environment (tests); `on_update(version, stats, engine)` runs on the learner thread after every update (bench)."""
    world_size, rank, local_rank, master_addr, master_port = rendezvous or (distributed_env() if args.distributed else (1, 0, 0, None, None))
    from . import topology
    topology.validate(args)
    lay = None
    if topology.is_split(args):  # actor and learner roles are different processes (README.md:58,62, benchmark.sh:80,90)
        if not args.distributed or world_size < 2:
            raise SystemExit("split topologies (--actor-device-ids != --learner-device-ids) run one process per role: launch through the "
                             "entry points (which fan out), torchrun or the SLURM variables, and pass --distributed")
        lay = topology.Layout(args, world_size, rank)
        n_proc, proc_index = lay.groups, lay.group   # the reference's world_size counts actor+learner groups (ppo:425-430)
    else:
        n_proc, proc_index = world_size, rank
    if getattr(args, "async_batch_size", 0) and algo != "ppo":
        raise SystemExit("--async-batch-size belongs to the PPO script (cleanba_impala.py already drives envpool through recv/send)")
    finalize(args, n_proc, proc_index)
    rdv = topology.Rendezvous(world_size, rank, master_addr, master_port) if world_size > 1 else None
    synthetic = args.env_backend != "envpool"
    run_name = f"{args.env_id}{'-synthetic' if synthetic else ''}__{args.exp_name}__{args.seed}__{uuid.uuid4()}"
    if rdv is not None:
        run_name = rdv.share("run_name", lambda: run_name.encode(), 0).decode()
    if synthetic and rank == 0:
        print(f"NOTE: --env-backend {args.env_backend} steps the built-in synthetic Atari-shaped env (84x84x4 uint8 frames, {args.num_actions} "
              f"actions), not ALE '{args.env_id}': returns are not comparable with the reference's {args.env_id} curves (the run name carries "
              "'-synthetic').  Use --env-backend envpool where envpool is installed.")
    if args.track and rank == 0:   # ppo:447-458 — same wandb.init call when wandb is importable; never a silent no-op
        try:
            import wandb
            wandb.init(project=args.wandb_project_name, entity=args.wandb_entity, sync_tensorboard=True, config=vars(args), name=run_name,
                       monitor_gym=True, save_code=True)
        except ImportError:
            print("--track: wandb is not installed here; scalars are written to TensorBoard event files under runs/ only")
    null_writer = SimpleNamespace(add_scalar=lambda *a: None, add_text=lambda *a: None, close=lambda: None)
    if rank == 0:
        writer = JsonlWriter(f"runs/{run_name}")
    elif lay is not None and lay.group == 0 and not lay.is_actor and lay.learner_index == 0:
        writer = JsonlWriter(f"runs/{run_name}/learner")   # the loss scalars of a split run come from learner 0 (rank 0 is an actor)
    else:
        writer = null_writer
    writer.add_text("hyperparameters", "|param|value|\n|-|-|\n%s" % ("\n".join([f"|{k}|{v}|" for k, v in vars(args).items()] +
                                                                            [f"|synthetic_env|{synthetic}|"])))

    # seeding (ppo:465-470): identical model / learner keys in every process, env seeds differ by process index
    key = prng.prng_key(args.seed)
    key, network_key, actor_key, critic_key = prng.split(key, 4)
    learner_key = key.copy()

    cfg = make_config(args, algo)
    if lay is not None:
        if engine_factory is None:
            # the id lists are relative to the group's visible devices (the reference's recipe gives every group its own HIP_VISIBLE_DEVICES,
            # README.md:71-72); a launcher that starts ALL groups with one common device set (bench.py --topology "2x(...)" on an 8-GPU node) says
            # how far apart the groups sit: CBM_GROUP_DEVICE_STRIDE
            cfg.device = lay.device_id + lay.group * int(os.environ.get("CBM_GROUP_DEVICE_STRIDE", "0") or 0)
    if os.environ.get("CBM_FORCE_DEVICE"):   # testing aid: every role on this GPU (the split path between processes on a one-GPU box)
        cfg.device = int(os.environ["CBM_FORCE_DEVICE"])
    if lay is not None:
        if lay.is_actor:
            cfg.num_actor_slots = lay.threads            # this actor GPU's own threads
        else:                                            # learner role: its slots are ingest ports for E/L env columns each
            cfg.local_num_envs, cfg.num_actor_slots = lay.shard_envs, lay.ports
    engine = engine_factory(cfg) if engine_factory else HipEngine(cfg)
    try:
        params = M.init_params(args.network, args.num_actions, network_key, actor_key, critic_key, hidden=int(args.hiddens[0]))
        engine.set_params(params)
        if lay is not None:
            return _train_split(args, algo, engine, lay, rdv, writer, key, rank, run_name, on_update)
        return _train_single(args, algo, engine, rdv, writer, key, learner_key, world_size, rank, run_name, on_update, engine_factory)
    except BaseException:
        if rdv is not None:
            rdv.abort()
        engine.abort()
        raise


def log_losses(writer, algo, stats, global_step):
    """losses/* are the means over epochs x minibatches (ppo:651-655 / impala:636-639 take .mean() of the per-step values)."""
    m = np.asarray(stats, np.float64).mean(axis=0)
    writer.add_scalar("losses/value_loss", float(m[2]), global_step)
    writer.add_scalar("losses/policy_loss", float(m[1]), global_step)
    writer.add_scalar("losses/entropy", float(m[3]), global_step)
    if algo == "ppo":
        writer.add_scalar("losses/approx_kl", float(m[4]), global_step)
    writer.add_scalar("losses/loss", float(m[0]), global_step)


def _train_single(args, algo, engine, rdv, writer, key, learner_key, world_size, rank, run_name, on_update, engine_factory):
    """a0-l0 (x world_size processes with --distributed: the reference's a0_l0_dN, README.md:103-108)."""
    from . import topology
    if rdv is not None or engine.wants_comm_at_world_one():
        topology.setup_learner_comm(engine, rdv, list(range(world_size)), rank)
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
    start = time.time()
    last_stats = None
    while True:
        learner_policy_version += 1
        t0 = time.time()
        try:
            engine.learner_wait()  # every actor slot's rollout (ppo:697-711)
        except Exception:
            if errors:
                raise errors[0]
            raise
        rollout_queue_get_time.append(time.time() - t0)
        training_time_start = time.time()
        lrs, bc1, bc2 = schedules(args, algo, opt_count, n_opt)
        want_stats = learner_policy_version % args.log_frequency == 0 or learner_policy_version >= args.num_updates
        # one call per update on one GPU and on N: with a learner communicator the library all-reduces every minibatch's gradient
        # (pmean, ppo:628) and the requested loss statistics (ppo:649-653) itself
        learner_key, stats = engine.learner_update(learner_key, lrs, bc1, bc2, want_stats)
        opt_count += n_opt
        if stats is not None:
            last_stats = stats
        global_step = learner_policy_version * args.local_batch_size * world_size
        if on_update:
            on_update(learner_policy_version, stats, engine)
        if learner_policy_version % args.log_frequency == 0 and stats is not None:
            writer.add_scalar("stats/rollout_queue_get_time", np.mean(rollout_queue_get_time), global_step)
            writer.add_scalar("stats/training_time", time.time() - training_time_start, global_step)
            print(global_step, f"learner_policy_version={learner_policy_version}, training time: {time.time() - training_time_start}s")
            writer.add_scalar("charts/learning_rate", float(lrs[-1]), global_step)
            log_losses(writer, algo, stats, global_step)
        if learner_policy_version >= args.num_updates:
            break
    engine.sync()
    elapsed = time.time() - start
    stop_event.set()
    engine.abort()   # actor threads blocked on a ring entry / parameter version that will never come return now
    for th in threads:
        th.join()
    result = {"updates": learner_policy_version, "global_step": learner_policy_version * args.local_batch_size * world_size,
              "elapsed_s": elapsed, "stats": last_stats, "params": engine.get_params(), "run_name": run_name}
    _unmap_then_close(engine, rdv)   # (unmap -> barrier -> free: the native all-reduce's peers still map this rank's gradient window)
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
                    repo_name = f"{args.env_id}{'-synthetic' if args.env_backend != 'envpool' else ''}-{args.exp_name}-seed{args.seed}"
                    push_to_hub(args, rets, f"{args.hf_entity}/{repo_name}" if args.hf_entity else repo_name, "PPO" if algo == "ppo" else "IMPALA",
                                f"runs/{run_name}", f"videos/{run_name}-eval", extra_dependencies=["cleanba_amd"])
                except ImportError:
                    print("--upload-model: cleanrl_utils.huggingface is not installed here; the model stays at", path)
    writer.close()
    engine.close()
    return result


def _train_split(args, algo, engine, lay, rdv, writer, key, rank, run_name, on_update=None):
    """One role process of an actor/learner-split run (cleanba_amd.topology).  Actor roles run the rollout threads and write shards into
    the learners' rings; learner roles ingest them, all-reduce gradients over every learner of every group, and learner 0 writes the
    parameters back into its group's actors."""
    from . import topology
    n_opt = args.num_minibatches * (args.update_epochs if algo == "ppo" else 1)
    start = time.time()
    if lay.is_actor:
        receiver = topology.ParamReceiver(engine, lay, rdv, args.num_updates)     # publishes this actor's parameter-buffer handles
        shipper = topology.ActorShipper(engine, lay, rdv, args, algo, args.num_updates)   # maps the learners' rings
        stop_event, errors, threads = threading.Event(), [], []
        dummy_writer = SimpleNamespace(add_scalar=lambda x, y, z: None)

        def on_error():   # a rollout thread died: the shipper must stop waiting for its commits, learners / receivers must stop waiting for us
            shipper.stop.set()
            rdv.abort()
        for slot in range(lay.threads):
            th = threading.Thread(target=rollout, args=(key.copy(), args, algo, engine, writer if slot == 0 else dummy_writer, slot, lay.groups,
                                                        lay.group, stop_event, errors, shipper.on_commit, lay.actor_index * lay.threads + slot,
                                                        on_error),
                                  daemon=True)
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
        engine.abort()   # like the reference's actors, each thread is one rollout past the last update; release it
        for th in threads:
            th.join()
        engine.sync()
        result = {"updates": args.num_updates, "elapsed_s": elapsed, "stats": None, "params": engine.get_actor_params(),
                  "run_name": run_name, "role": "actor" if lay.na == 1 else f"actor{lay.actor_index}"}
        rdv.barrier("done")       # keep this actor's buffers mapped until every peer has stopped writing into them
        _unmap_then_close(engine, rdv)
        writer.close()
        return result
    ingest = topology.LearnerReceiver(engine, lay, rdv, args, algo, args.num_updates)   # publishes this learner's ring handles
    send_params = topology.ParamSender(engine, lay, rdv) if lay.learner_index == 0 else None
    topology.setup_learner_comm(engine, rdv, lay.all_learner_ranks, rank)
    ingest.thread.start()
    learner_key = key.copy()
    opt_count, stats = 0, None
    for version in range(1, args.num_updates + 1):
        try:
            engine.learner_wait()
        except Exception:
            if ingest.error:
                raise ingest.error
            raise
        t0 = time.time()
        lrs, bc1, bc2 = schedules(args, algo, opt_count, n_opt)
        learner_key, stats = engine.learner_update(learner_key, lrs, bc1, bc2, True)
        opt_count += n_opt
        if send_params is not None:
            send_params(version)
        if on_update:
            on_update(version, stats, engine)
        if version % args.log_frequency == 0 and lay.learner_index == 0:
            global_step = version * args.local_batch_size * lay.groups
            print(global_step, f"learner_policy_version={version}, training time: {time.time() - t0}s")
            writer.add_scalar("stats/training_time", time.time() - t0, global_step)
            writer.add_scalar("charts/learning_rate", float(lrs[-1]), global_step)
            log_losses(writer, algo, stats, global_step)
    engine.sync()
    ingest.thread.join()
    if ingest.error:
        raise ingest.error
    result = {"updates": args.num_updates, "elapsed_s": time.time() - start, "stats": stats, "params": engine.get_params(), "run_name": run_name,
              "role": f"learner{lay.learner_index}"}
    rdv.barrier("done")
    _unmap_then_close(engine, rdv)
    writer.close()
    return result


def _unmap_then_close(engine, rdv, tag="unmapped"):
    """Teardown order of a run that shares device buffers between processes: every process unmaps what it mapped of its peers, a barrier, and
    only then does an owner free (engine.close) — no owner frees a window a peer still maps."""
    if hasattr(engine, "unmap_peers"):
        engine.unmap_peers()
    if rdv is not None:
        rdv.barrier(tag)
    engine.close()


class HipEngine(L.Context):
    """The product engine: cleanba_amd.lib.Context plus the handle plumbing of the multi-process paths (no torch tensors, no CPU fallback)."""

    def wants_comm_at_world_one(self):
        """CBM_FORCE_DIST=1: build a real one-rank RCCL communicator so the all-reduce path runs on a single GPU (tests, bench)."""
        return os.environ.get("CBM_FORCE_DIST") == "1"

    def comm_unique_id(self):
        if L._TORCH_RCCL:
            L._chk(self.lib.cbm_comm_load(L._TORCH_RCCL.encode()))
        buf = (L.C.c_uint8 * L.COMM_ID_BYTES)()
        L._chk(self.lib.cbm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, uid, nranks, rank, which=L.COMM_LEARNERS):
        super().comm_init(which, uid, nranks, rank)

    # ---- split topologies: the context's export window 0 (ring fields + versioned actor parameter buffers) and the peers' mappings of it:
    # ONE HIP IPC mapping per peer context, fields addressed by offset (include/cleanba_mi.h, "EXPORT WINDOWS")
    def export_ring(self, fields, tag=-1):
        return {"cols": self.cfg.local_num_envs * self.cfg.num_actor_slots, "window": self.ipc_export_window(0, tag), "tag": tag, "pid": os.getpid(),
                "entries": [{f: self.ipc_window_offset(f, r)[1] for f in fields} for r in range(self.cfg.ring_depth)]}

    def open_peer_ring(self, desc, what=""):
        base = self.ipc_open_window(desc["window"], what or f"ring of exporter tag {desc.get('tag')} (pid {desc.get('pid')})")
        out = []
        for entry in desc["entries"]:
            pr = L.PeerRing()
            for f, off in entry.items():
                setattr(pr, f, base + off)
            out.append(pr)
        return out

    def export_actor_params(self, tag=-1):
        return {"window": self.ipc_export_window(0, tag), "tag": tag, "pid": os.getpid(),
                "offsets": [self.ipc_window_offset(f"actor_params_v{i}")[1] for i in range(3)]}

    def open_peer_params(self, desc, what=""):
        base = self.ipc_open_window(desc["window"], what or f"actor parameter buffers of exporter tag {desc.get('tag')} (pid {desc.get('pid')})")
        return [base + off for off in desc["offsets"]]

    def get_actor_params(self):
        return self.read("actor_params_latest", np.float32)
