"""Worker for the split-topology tests: one rank of a run with 1 actor + L learner processes per group on CPU (gloo + OracleEngine).
Usage: python split_worker.py <rank> <world> <port> <out.npz> <algo> <num_learners>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

rank, world, port, out, algo, nl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6])
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))

from cleanba_amd.args import parse_args  # noqa: E402
from cleanba_amd.trainer import train  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402

groups = world // (1 + nl)
E = 4 * nl
argv = ["--local-num-envs", str(E), "--num-actor-threads", "1", "--num-steps", "4", "--env-backend", "host", "--total-timesteps",
        str(3 * E * 4 * groups), "--log-frequency", "1", "--update-epochs", "1", "--network", "nature", "--distributed",
        "--actor-device-ids", "0", "--learner-device-ids"] + [str(i + 1) for i in range(nl)]
args = parse_args(argv, algo)
os.chdir(os.environ.get("CBM_TEST_TMP", "/tmp"))
res = train(args, algo, engine_factory=OracleEngine)
np.savez(out, params=res["params"], role=np.array(res["role"]), updates=res["updates"])
print("rank", rank, res["role"], "updates", res["updates"])
