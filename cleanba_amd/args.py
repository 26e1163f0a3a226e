"""CLI surface of cleanba_ppo.py / cleanba_impala.py (Args dataclass ppo:34-118, impala:34-110).

tyro is not installable here, so the same flags are parsed with argparse: `--local-num-envs` and
`--local_num_envs` are both accepted (README.md:56 vs benchmark.sh:80), bools are `--x/--no-x`
(`--no-concurrency`, benchmark.sh:26), list flags are space separated (`--learner-device-ids 1 2 3`,
README.md:62).  Build-only additions are marked [mi].
"""
import argparse
import os
from dataclasses import dataclass, field, fields
from typing import List, Optional


@dataclass
class Args:
    exp_name: str = "cleanba_ppo"
    seed: int = 1
    track: bool = False
    wandb_project_name: str = "cleanRL"
    wandb_entity: Optional[str] = None
    capture_video: bool = False
    save_model: bool = False
    upload_model: bool = False
    hf_entity: str = ""
    log_frequency: int = 10

    env_id: str = "Breakout-v5"
    total_timesteps: int = 50000000
    learning_rate: float = 2.5e-4
    local_num_envs: int = 64
    num_actor_threads: int = 2
    num_steps: int = 128
    anneal_lr: bool = True
    gamma: float = 0.99
    gae_lambda: float = 0.95
    num_minibatches: int = 4
    gradient_accumulation_steps: int = 1
    update_epochs: int = 4
    norm_adv: bool = True
    clip_coef: float = 0.1
    ent_coef: float = 0.01
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    channels: List[int] = field(default_factory=lambda: [16, 32, 32])
    hiddens: List[int] = field(default_factory=lambda: [256])

    actor_device_ids: List[int] = field(default_factory=lambda: [0])
    learner_device_ids: List[int] = field(default_factory=lambda: [0])
    distributed: bool = False
    concurrency: bool = False

    # [mi] build-only
    network: str = "impala_resnet"   # "impala_resnet" (ppo:149-189, the reference default) | "nature" (naturecnn:143-178)
    env_backend: str = "device"      # "device": synthetic env stepping on the GPU; "host": same env on the CPU
                                     # through the envpool API; "envpool": real envpool if installed
    num_actions: int = 18            # full_action_space=True (ppo:135)
    max_updates: int = 0             # stop early after this many updates (0 = total_timesteps)
    eval_episodes: int = 10  # episodes of the post-training evaluation that follows --save-model (ppo:773-782)
    eval_max_episode_steps: int = 0  # 0 = the env's own limit (27000 for Atari); smaller values bound smoke runs
    async_batch_size: int = 0  # legacy `--async-batch-size` (legacy_scripts/..._naturecnn.py:65-66): envpool async mode, recv() returns this many
                               # of the local_num_envs envs; env-id-indexed GAE, per-minibatch advantage normalisation.  0 = off (cleanba_ppo.py)
    backward_split: int = 0  # build-only extension: 0 = backward GEMMs on fp32 MFMA (reference precision); 2 / 3 = fp32 operands split exactly into
                             # 2 / 3 bf16 terms, products on bf16 MFMA, fp32 accumulate (gradient error ~1e-6 / ~1e-7 of the fp32 path's); Nature-CNN
    conv1_fp32_chain: int = 0  # Nature-CNN, learner minibatches (> 512 frames): 0 = conv1 forward / weight gradient as exact uint8 x three-term-bf16 products on the
                               # bf16 matrix cores, fp32 accumulate (default; logits within 1e-6 of the chain); 3 = both as fp32-MFMA fmaf chains, bit-identical to the
                               # CPU oracle (rounds 1-5); 1 / 2 = only the forward / only the weight gradient on the chain
    bf16_forward: bool = False  # build-only extension (reference is fp32): conv2/conv3/dense forward on bf16 MFMA, fp32 accumulate + fp32 returns (Nature-CNN)
    same_env_seed_all_ranks: bool = False  # testing aid: every process steps identical envs (then dp-N == dp-1 bitwise)

    # runtime arguments to be filled in (ppo:105-117)
    local_batch_size: int = 0
    local_minibatch_size: int = 0
    num_updates: int = 0
    world_size: int = 0
    local_rank: int = 0
    num_envs: int = 0
    batch_size: int = 0
    minibatch_size: int = 0
    global_learner_decices: Optional[List[str]] = None
    actor_devices: Optional[List[str]] = None
    learner_devices: Optional[List[str]] = None


IMPALA_OVERRIDES = dict(exp_name="cleanba_impala", learning_rate=0.0006, num_steps=20, max_grad_norm=40.0, concurrency=True,
                        update_epochs=1, norm_adv=False)
_RUNTIME = {"local_batch_size", "local_minibatch_size", "num_updates", "world_size", "local_rank", "num_envs", "batch_size",
            "minibatch_size", "global_learner_decices", "actor_devices", "learner_devices"}


def build_parser(algo="ppo"):
    defaults = Args()
    if algo == "impala":
        for k, v in IMPALA_OVERRIDES.items():
            setattr(defaults, k, v)
    p = argparse.ArgumentParser(prog=f"cleanba_{algo}", allow_abbrev=False)
    for f in fields(Args):
        if f.name in _RUNTIME:
            continue
        d = getattr(defaults, f.name)
        names = ["--" + f.name.replace("_", "-")]
        if "_" in f.name:
            names.append("--" + f.name)
        if isinstance(d, bool):
            p.add_argument(*names, dest=f.name, action="store_true", default=d)
            p.add_argument(*["--no-" + n[2:] for n in names], dest=f.name, action="store_false")
        elif isinstance(d, list):
            p.add_argument(*names, dest=f.name, type=int, nargs="+", default=list(d))
        elif d is None:
            p.add_argument(*names, dest=f.name, type=str, default=None)
        else:
            p.add_argument(*names, dest=f.name, type=type(d), default=d)
    return p


def parse_args(argv=None, algo="ppo"):
    ns = build_parser(algo).parse_args(argv)
    args = Args()
    for k, v in vars(ns).items():
        setattr(args, k, v)
    return args


def finalize(args, world_size=1, rank=0):
    """Derived fields and the reference's assertions (ppo:411-430)."""
    n_actor_dev, n_learner = len(args.actor_device_ids), len(args.learner_device_ids)
    # --channels / --hiddens (ppo:92-95) travel in cbm_config; cbm_ctx_create takes --hiddens H (one layer, 64..512 step 64) and the default
    # --channels only, and rejects other widths with the same message for a C host as for this CLI (trainer.make_config)
    if getattr(args, "async_batch_size", 0):
        # the legacy async script (naturecnn:102-105): one actor thread on one actor device, host envs, PPO, learner on the same GPU
        if args.local_num_envs % args.async_batch_size:
            raise SystemExit("--local-num-envs must be a multiple of --async-batch-size (naturecnn:104)")
        if args.env_backend == "device":
            raise SystemExit("--async-batch-size models envpool's async recv/send: use --env-backend host (or envpool)")
        if n_actor_dev != 1 or n_learner != 1 or args.actor_device_ids != args.learner_device_ids or args.gradient_accumulation_steps != 1:
            raise SystemExit("--async-batch-size runs a0-l0 without gradient accumulation (naturecnn:105)")
        args.num_actor_threads = 1
        if (args.local_num_envs * args.num_steps) % args.num_minibatches:
            raise SystemExit("local_num_envs*num_steps must be divisible by num_minibatches")
        args.local_batch_size = int(args.local_num_envs * args.num_steps)
        args.local_minibatch_size = int(args.local_batch_size // args.num_minibatches)
        args.world_size, args.local_rank = world_size, rank
        args.num_envs = args.local_num_envs * world_size
        args.batch_size = args.local_batch_size * world_size
        args.minibatch_size = args.local_minibatch_size * world_size
        args.num_updates = args.total_timesteps // (args.local_batch_size * world_size)
        if args.max_updates:
            args.num_updates = min(args.num_updates, args.max_updates)
        return args
    args.local_batch_size = int(args.local_num_envs * args.num_steps * args.num_actor_threads * n_actor_dev)
    args.local_minibatch_size = int(args.local_batch_size // args.num_minibatches)
    assert args.local_num_envs % n_learner == 0, "local_num_envs must be divisible by len(learner_device_ids)"
    assert int(args.local_num_envs / n_learner) * args.num_actor_threads % args.num_minibatches == 0, \
        "int(local_num_envs / len(learner_device_ids)) must be divisible by num_minibatches"
    args.world_size = world_size
    args.local_rank = rank
    args.num_envs = args.local_num_envs * world_size * args.num_actor_threads * n_actor_dev
    args.batch_size = args.local_batch_size * world_size
    args.minibatch_size = args.local_minibatch_size * world_size
    args.num_updates = args.total_timesteps // (args.local_batch_size * world_size)
    if args.max_updates:
        args.num_updates = min(args.num_updates, args.max_updates)
    return args


def distributed_env():
    """Process rendezvous contract: torchrun's RANK/WORLD_SIZE, or the reference's fake-SLURM variables
    (README.md:71-72: SLURM_NTASKS / SLURM_PROCID / SLURM_LOCALID / SLURM_STEP_NODELIST)."""
    e = os.environ
    if "WORLD_SIZE" in e:
        return int(e["WORLD_SIZE"]), int(e.get("RANK", 0)), int(e.get("LOCAL_RANK", 0)), e.get("MASTER_ADDR", "127.0.0.1"), int(
            e.get("MASTER_PORT", 29500))
    if "SLURM_NTASKS" in e:
        host = e.get("SLURM_STEP_NODELIST", "localhost").split(",")[0]
        host = "127.0.0.1" if host == "localhost" else host
        return int(e["SLURM_NTASKS"]), int(e.get("SLURM_PROCID", 0)), int(e.get("SLURM_LOCALID", 0)), host, 29500 + int(
            e.get("SLURM_JOB_ID", 0)) % 1000
    return 1, 0, 0, "127.0.0.1", 29500
