"""Seed x env-id fan-out of training commands — counterpart of `python -m cleanrl_utils.benchmark` as the reference's README.md:74-83 and
benchmark.sh use it (cleanrl_utils/benchmark.py:12-137): same flags, same command matrix (`<command> --env-id E --seed S`, seeds outer,
env ids inner), local execution on `--workers` parallel workers, or a SLURM array script rendered from a template with the reference's
`{{placeholder}}` names (cleanba.slurm_template) and submitted with sbatch.

ROCm notes: the rendered script is whatever the template says — give it an MI355X template (`--gpus-per-task`, `module load rocm`, the
`srun` line); inside a task, `cleanba_amd.cleanba_ppo --distributed` reads SLURM_NTASKS / SLURM_PROCID / SLURM_LOCALID exactly like the
reference (cleanba_amd.args.distributed_env).  The wandb auto-tag of the reference needs git tags + network and is not reproduced.
"""
import argparse
import math
import os
import shlex
import subprocess
import uuid
from concurrent.futures import ThreadPoolExecutor


def build_parser():
    p = argparse.ArgumentParser(prog="cleanba_amd.benchmark")
    p.add_argument("--env-ids", nargs="+", default=["Breakout-v5"])
    p.add_argument("--command", type=str, default="python -m cleanba_amd.cleanba_ppo")
    p.add_argument("--num-seeds", type=int, default=3)
    p.add_argument("--start-seed", type=int, default=1)
    p.add_argument("--workers", type=int, default=0, help="0 = only print the commands (and write the SLURM script, if a template is given)")
    p.add_argument("--slurm-template-path", type=str, default=None)
    p.add_argument("--slurm-gpus-per-task", type=int, default=1)
    p.add_argument("--slurm-total-cpus", type=int, default=50)
    p.add_argument("--slurm-ntasks", type=int, default=1)
    p.add_argument("--slurm-nodes", type=int, default=None)
    return p


def command_matrix(command, env_ids, num_seeds, start_seed=1):
    return [f"{command} --env-id {env_id} --seed {start_seed + s}" for s in range(num_seeds) for env_id in env_ids]


def render_slurm(template, args, n_commands):
    """Fills the reference template's placeholders.  Array task i runs env_ids[i / len_seeds] with seeds[i % len_seeds]."""
    seeds = [str(args.start_seed + s) for s in range(args.num_seeds)]
    cpus_per_gpu = math.ceil(args.slurm_total_cpus / (args.slurm_gpus_per_task * args.slurm_ntasks))
    fill = {"array": f"0-{n_commands - 1}%{args.workers}", "env_ids": "(" + " ".join(args.env_ids) + ")", "seeds": "(" + " ".join(seeds) + ")",
            "len_seeds": str(args.num_seeds), "command": args.command, "gpus_per_task": str(args.slurm_gpus_per_task),
            "cpus_per_gpu": str(cpus_per_gpu), "ntasks": str(args.slurm_ntasks),
            "nodes": f"#SBATCH --nodes={args.slurm_nodes}" if args.slurm_nodes is not None else ""}
    for k, v in fill.items():
        template = template.replace("{{" + k + "}}", v)
    return template


def _run(command):
    print(f"running {command}", flush=True)
    rc = subprocess.call(shlex.split(command))
    if rc != 0:
        raise RuntimeError(f"`{command}` exited with {rc}")
    return rc


def main(argv=None):
    args = build_parser().parse_args(argv)
    commands = command_matrix(args.command, args.env_ids, args.num_seeds, args.start_seed)
    print("======= commands to run:")
    for c in commands:
        print(c)
    if args.slurm_template_path is None:
        if args.workers <= 0:
            print("not running the experiments because --workers is set to 0; just printing the commands to run")
            return commands
        with ThreadPoolExecutor(max_workers=args.workers, thread_name_prefix="cleanba-benchmark-worker-") as pool:
            for f in [pool.submit(_run, c) for c in commands]:
                f.result()
        return commands
    os.makedirs(os.path.join("slurm", "logs"), exist_ok=True)
    with open(args.slurm_template_path) as f:
        script = render_slurm(f.read(), args, len(commands))
    path = os.path.join("slurm", f"{uuid.uuid4()}.slurm")
    with open(path, "w") as f:
        f.write(script)
    print(f"saving command in {path}")
    if args.workers > 0:
        _run(f"sbatch {path}")
    return path


if __name__ == "__main__":
    main()
