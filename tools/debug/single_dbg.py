import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from cleanba_amd.args import parse_args
from cleanba_amd.trainer import train
os.chdir("/tmp")
algo = sys.argv[1]
E, T, updates = 8, 8, 3
base = ["--local-num-envs", str(E), "--num-actor-threads", "2", "--num-steps", str(T), "--env-backend", "device", "--network", "nature",
        "--total-timesteps", str(updates * E * 2 * T), "--log-frequency", "1000", "--update-epochs", "1"]
ref = None
for i in range(int(os.environ.get("REPS", "10"))):
    r = train(parse_args(base, algo), algo)["params"]
    if ref is None: ref = r
    print(algo, "rep", i, "same as first", np.array_equal(r, ref), flush=True)
