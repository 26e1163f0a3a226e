"""Worker for tests/test_distributed_cpu.py: one rank of a world_size-N data-parallel run on CPU (gloo).
Usage: python dist_worker.py <rank> <world> <port> <out.npy> <same_seed 0/1> [algo]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

rank, world, port, out, same = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
algo = sys.argv[6] if len(sys.argv) > 6 else "ppo"
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))

from cleanba_amd.args import parse_args  # noqa: E402
from cleanba_amd.trainer import train  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402

argv = ["--local-num-envs", "4", "--num-actor-threads", "1", "--num-steps", "4", "--env-backend", "host", "--total-timesteps",
        str(int(os.environ.get("CBM_TEST_UPDATES", "2")) * 4 * 4 * world), "--log-frequency", "1", "--update-epochs", "1", "--network", "nature"]
if world > 1:
    argv.append("--distributed")
if same:
    argv.append("--same-env-seed-all-ranks")
if os.environ.get("CBM_TEST_ACCUM"):
    argv += ["--gradient-accumulation-steps", os.environ["CBM_TEST_ACCUM"], "--num-minibatches", "2"]
if os.environ.get("CBM_TEST_ASYNC"):   # legacy envpool async mode: recv() batches of 2 of the 4 envs
    argv += ["--async-batch-size", os.environ["CBM_TEST_ASYNC"]]
args = parse_args(argv, algo)
os.chdir(os.environ.get("CBM_TEST_TMP", "/tmp"))
res = train(args, algo, engine_factory=OracleEngine)
np.save(out, res["params"])
print("rank", rank, "updates", res["updates"], "stats", None if res["stats"] is None else res["stats"][-1])
