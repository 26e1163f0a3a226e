"""Worker for tests/test_gpu_split.py: one ROLE process (actor or learner) of a split run on the HIP engine.  All roles are pinned to GPU 0
(CBM_FORCE_DEVICE) so the IPC peer-write path — handle export / open, strided shard copies, parameter push, 'landed' messages — runs between
real processes on a one-GPU box.  Usage: python gpu_split_worker.py <rank> <world> <port> <out.npz> <algo> <E> <T> <updates> <threads>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

rank, world, port, out, algo = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
E, T, updates, threads = (int(x) for x in sys.argv[6:10])
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CBM_FORCE_DEVICE="0")

from cleanba_amd.args import parse_args  # noqa: E402
from cleanba_amd.trainer import train  # noqa: E402

argv = ["--local-num-envs", str(E), "--num-actor-threads", str(threads), "--num-steps", str(T), "--env-backend", "device", "--network", "nature",
        "--total-timesteps", str(updates * E * threads * T), "--log-frequency", "1000", "--update-epochs", "1", "--distributed",
        "--actor-device-ids", "0", "--learner-device-ids", "1"]
os.chdir(os.environ.get("CBM_TEST_TMP", "/tmp"))
res = train(parse_args(argv, algo), algo)
np.savez(out, params=res["params"], role=np.array(res["role"]), updates=res["updates"])
print("rank", rank, res["role"], "updates", res["updates"])
