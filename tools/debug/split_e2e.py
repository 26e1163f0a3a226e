"""Deviation of whole PPO runs (HIP engine, backward_split 0/2/3) from the fp32 oracle engine run: max |param diff| after U updates."""
import os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from oracle_engine import OracleEngine
from cleanba_amd.args import parse_args
from cleanba_amd.trainer import train
os.chdir(tempfile.mkdtemp())
for (E, thr, T, nmb, ep) in ((4, 2, 5, 2, 2), (8, 1, 16, 4, 4)):
    updates = 4
    argv = ["--local-num-envs", str(E), "--num-actor-threads", str(thr), "--num-steps", str(T), "--num-minibatches", str(nmb), "--update-epochs", str(ep),
            "--network", "nature", "--env-backend", "host", "--total-timesteps", str(updates * E * thr * T), "--log-frequency", "1000", "--concurrency"]
    cpu = train(parse_args(argv, "ppo"), "ppo", engine_factory=OracleEngine)
    for split in (0, 2, 3):
        gpu = train(parse_args(argv + ["--backward-split", str(split)], "ppo"), "ppo")
        d = np.abs(gpu["params"] - cpu["params"])
        print(f"E={E} thr={thr} T={T}: split {split}: max |dparam| {d.max():.3e}  (max |param| {np.abs(cpu['params']).max():.3f}, moved {np.abs(cpu['params'] - cpu['params0']).max() if 'params0' in cpu else float('nan'):.3e})  stats max rel {np.max(np.abs(gpu['stats'] - cpu['stats']) / (np.abs(cpu['stats']) + 1e-6)):.2e}", flush=True)
