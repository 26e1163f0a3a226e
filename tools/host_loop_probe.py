"""Per-step breakdown of the envpool-API path INSIDE the product trainer (concurrent learner included): the trainer's own timers
(stats/inference_time = cbm_actor_step_host, stats/env_send_time = envs.step, stats/storage_time = record + bookkeeping), per 120-env step.
usage (GPU box): python tools/host_loop_probe.py [threads]"""
import os
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanba_amd.args import parse_args  # noqa: E402
from cleanba_amd import trainer  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 1
E, T, warm, n_up = 120, 128, 2, 8
acc = defaultdict(list)


class W:
    def add_scalar(self, tag, val, step):
        acc[tag].append(float(val))

    def add_text(self, *a):
        pass

    def close(self):
        pass


marks = {}


def on_update(v, stats, e):
    if v == warm or v == warm + n_up:
        e.sync()
        marks[v] = time.perf_counter()


argv = ["--local-num-envs", str(E // threads), "--num-actor-threads", str(threads), "--num-steps", str(T), "--env-backend", "host", "--network", "nature",
        "--total-timesteps", str((warm + n_up) * E * T), "--log-frequency", "1", "--concurrency"]
os.chdir(os.environ.get("TMPDIR", "/tmp"))
trainer.JsonlWriter = lambda logdir: W()
so = sys.stdout
sys.stdout = open(os.devnull, "w")
try:
    trainer.train(parse_args(argv, "ppo"), "ppo", on_update=on_update)
finally:
    sys.stdout = so
dt = marks[warm + n_up] - marks[warm]
print("threads %d: %.1f k env-steps/s, %.1f us per %d-env step (wall)" % (threads, n_up * E * T / dt / 1e3, dt / (n_up * T) * 1e6 / 1, E))
for tag in ("stats/inference_time", "stats/env_send_time", "stats/storage_time", "stats/rollout_time", "stats/params_queue_get_time", "stats/rollout_queue_put_time"):
    if acc[tag]:
        v = np.array(acc[tag][2:])
        per = v.mean() / (T if "queue" not in tag else 1) * 1e6
        print("  %-32s %8.1f us per %s" % (tag, per, "step" if "queue" not in tag else "rollout"))
