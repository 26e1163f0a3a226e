"""Runs three updates of a handful of off-headline configurations through the product trainer (two actor threads = 7680-frame minibatches, IMPALA-ResNet
T = 20, 64 x 64 and 40 x 128 batches) and prints whether the parameters stayed finite: a crash / hang check for batch-size-dependent kernel paths
(frames per block, ring sizes), not a parity test.  usage (GPU box): python tools/robust_run.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanba_amd.args import parse_args
from cleanba_amd.trainer import train
os.chdir("/tmp")
for algo, extra in (("ppo", ["--network", "impala_resnet", "--num-actor-threads", "2", "--local-num-envs", "120"]),
                    ("impala", ["--network", "impala_resnet", "--num-steps", "20", "--local-num-envs", "120"]),
                    ("ppo", ["--network", "nature", "--num-actor-threads", "2", "--local-num-envs", "120"]),
                    ("ppo", ["--network", "impala_resnet", "--local-num-envs", "64", "--num-steps", "64"]),
                    ("ppo", ["--network", "nature", "--local-num-envs", "40", "--num-steps", "128"])):
    a = parse_args(extra + ["--env-backend", "device", "--total-timesteps", "1", "--log-frequency", "100000"], algo)
    per = a.local_num_envs * a.num_actor_threads * a.num_steps
    a = parse_args(extra + ["--env-backend", "device", "--total-timesteps", str(3 * per), "--log-frequency", "100000"], algo)
    so = sys.stdout; sys.stdout = open(os.devnull, "w")
    try:
        r = train(a, algo)
    finally:
        sys.stdout = so
    print(algo, extra, "updates", r["updates"], "finite", bool(np.isfinite(r["params"]).all()), "|p|", float(np.abs(r["params"]).sum()))
