"""Re-measures README.md's table of secondary workloads on the GPU box through the product trainer (cleanba_amd.trainer.train): device time
between the completion of update 3 and of update 15 (two syncs per run); the run goes on for three more updates, because its LAST updates have no
rollout beside them (the actor is done) and are 5-15 % faster than a steady-state one — round 6 found the table 3-4 % above bench.py's secondary
rows for that reason.  usage: python tools/readme_table.py [substring filter]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanba_amd.args import parse_args  # noqa: E402
from cleanba_amd.trainer import train  # noqa: E402

E, T = 120, 128
ROWS = [
    ("PPO nature T=128", "ppo", ["--network", "nature", "--env-backend", "device"], 1, T),
    ("PPO resnet T=128", "ppo", ["--network", "impala_resnet", "--env-backend", "device"], 1, T),
    ("IMPALA nature T=128", "impala", ["--network", "nature", "--env-backend", "device"], 1, T),
    ("IMPALA nature T=128 bf16 forward (configs[2])", "impala", ["--network", "nature", "--env-backend", "device", "--bf16-forward"], 1, T),
    ("IMPALA nature T=128 two actor threads", "impala", ["--network", "nature", "--env-backend", "device"], 2, T),
    ("IMPALA resnet T=128", "impala", ["--network", "impala_resnet", "--env-backend", "device"], 1, T),
    ("PPO nature T=128 two actor threads (240 envs)", "ppo", ["--network", "nature", "--env-backend", "device"], 2, T),
    ("IMPALA nature T=20 two actor threads", "impala", ["--network", "nature", "--env-backend", "device"], 2, 20),
    ("PPO nature T=128 two actor threads x 60 envs", "ppo", ["--network", "nature", "--env-backend", "device"], 2, T, 60),
    ("PPO nature T=128 four actor threads x 30 envs", "ppo", ["--network", "nature", "--env-backend", "device"], 4, T, 30),
    ("PPO nature T=128 backward-split 2", "ppo", ["--network", "nature", "--env-backend", "device", "--backward-split", "2"], 1, T),
    ("PPO nature host env 1 thread", "ppo", ["--network", "nature", "--env-backend", "host"], 1, T),
    ("PPO nature Atari57 mix", "ppo", ["--network", "nature", "--env-backend", "device", "--env-id", "Atari57Mix-v5"], 1, T),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
os.chdir(os.environ.get("TMPDIR", "/tmp"))
for name, algo, extra, threads, t, *rest in ROWS:
    E = rest[0] if rest else 120
    if flt not in name:
        continue
    warm, n_up = 3, (12 if "resnet" not in name else 6)
    if t < 64:
        warm, n_up = 40, 80     # short updates (4-5 ms): a 3-update warm-up still sits inside the first process's clock ramp / code-object loading
    total = warm + n_up
    marks = {}

    def on_update(v, st, e, marks=marks, warm=warm, total=total):
        if v == warm or v == total:      # two syncs per run: the learner thread keeps enqueueing ahead in between, like the product run
            e.sync()
            marks[v] = time.perf_counter()

    argv = ["--local-num-envs", str(E), "--num-actor-threads", str(threads), "--num-steps", str(t), "--total-timesteps",
            str((total + 3) * E * threads * t), "--log-frequency", "100000", "--concurrency"] + extra
    so = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        train(parse_args(argv, algo), algo, on_update=on_update)
    except BaseException as e:  # noqa: BLE001
        sys.stdout = so
        print("%-50s FAILED %r" % (name, e))
        continue
    finally:
        sys.stdout = so
    if warm not in marks or total not in marks:
        print("%-50s versions seen: %s" % (name, sorted(marks)))
        continue
    dt = (marks[total] - marks[warm]) / n_up
    per = E * threads * t
    print("%-50s %9.1f k env-steps/s   %7.2f ms per update (%d envs x %d steps)" % (name, per / dt / 1e3, dt * 1e3, E * threads, t))
