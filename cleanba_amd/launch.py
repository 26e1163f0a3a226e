"""Process fan-out for split actor/learner topologies, so that the reference's own command lines work unchanged.

The reference runs `--actor-device-ids 0 --learner-device-ids 1 2 3` inside ONE JAX process (README.md:62) and scales out with one such
process per group (`--distributed` + the SLURM variables, README.md:71-72).  Here every ROLE is its own process (cleanba_amd.topology), so
the entry points call `maybe_fan_out`: a process that was asked for a split topology and is not already a role worker spawns its group's
G = len(actor ids) + len(learner ids) workers — RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in torchrun's convention, LOCAL_RANK = the GPU
index from the id lists (relative to HIP_/CUDA_VISIBLE_DEVICES, like the reference) — and waits for them.  With `--distributed` the parent's
SLURM task index selects the group: global rank = SLURM_PROCID * G + position, world = SLURM_NTASKS * G.
"""
import os
import subprocess
import sys

from . import topology


def plan(args, environ):
    """Environment dicts of the workers this process must spawn, or None when it is a worker itself / no split was requested."""
    topology.validate(args)      # unsupported id lists fail here, before anything is spawned
    if not topology.is_split(args):
        return None
    if "RANK" in environ and "WORLD_SIZE" in environ:      # started by torchrun or by the fan-out below
        return None
    ids = list(args.actor_device_ids) + list(args.learner_device_ids)
    G = len(ids)
    if args.distributed and "SLURM_NTASKS" in environ:
        groups, group = int(environ["SLURM_NTASKS"]), int(environ.get("SLURM_PROCID", 0))
        host = environ.get("SLURM_STEP_NODELIST", "localhost").split(",")[0]
        host = "127.0.0.1" if host == "localhost" else host
        port = 29500 + int(environ.get("SLURM_JOB_ID", 0)) % 1000
    else:
        groups, group, host = 1, 0, "127.0.0.1"
        port = 29500 + os.getpid() % 1000
    out = []
    for pos, dev in enumerate(ids):
        e = dict(environ)
        e.update(RANK=str(group * G + pos), WORLD_SIZE=str(groups * G), LOCAL_RANK=str(dev), MASTER_ADDR=host, MASTER_PORT=str(port),
                 HSA_ENABLE_IPC_MODE_LEGACY=environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out.append(e)
    return out


def maybe_fan_out(args, module, argv):
    """Returns None when the caller should simply continue (worker or plain run); otherwise runs the group's workers and returns their
    worst exit code."""
    envs = plan(args, os.environ)
    if envs is None:
        return None
    cmd = [sys.executable, "-m", module] + list(sys.argv[1:] if argv is None else argv)
    if "--distributed" not in cmd:
        cmd.append("--distributed")          # the workers always rendezvous (cleanba_amd.topology.Rendezvous)
    procs = [subprocess.Popen(cmd, env=e) for e in envs]
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)
