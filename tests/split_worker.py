"""Worker for the split-topology tests: one rank of a run with 1 actor + L learner processes per group on CPU (gloo + OracleEngine).
Usage: python split_worker.py <rank> <world> <port> <out.npz> <algo> <num_learners | actor_ids:learner_ids e.g. 0:0,1>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

rank, world, port, out, algo = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
if ":" in sys.argv[6]:
    aids, lids = (x.split(",") for x in sys.argv[6].split(":"))
else:
    aids, lids = ["0"], [str(i + 1) for i in range(int(sys.argv[6]))]
nl = len(lids)
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))

from cleanba_amd.args import parse_args  # noqa: E402
from cleanba_amd.trainer import train  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402

groups = world // (len(aids) + nl)
E = 4 * nl
argv = ["--local-num-envs", str(E), "--num-actor-threads", "1", "--num-steps", "4", "--env-backend", "host", "--total-timesteps",
        str(3 * E * 4 * groups * len(aids)), "--log-frequency", "1", "--update-epochs", "1", "--network", "nature", "--distributed",
        "--actor-device-ids"] + aids + ["--learner-device-ids"] + lids
args = parse_args(argv, algo)
os.chdir(os.environ.get("CBM_TEST_TMP", "/tmp"))
factory = OracleEngine
fail_at = int(os.environ.get("CBM_TEST_FAIL_ROLLOUT", "0"))
if fail_at:
    # failure injection: the actor's rollout thread dies when it begins rollout `fail_at`; every role must stop promptly
    class FailingEngine(OracleEngine):
        _begun = 0

        def actor_begin_rollout(self, *a, **k):
            FailingEngine._begun += 1
            if FailingEngine._begun == fail_at:
                raise RuntimeError("injected rollout failure")
            return super().actor_begin_rollout(*a, **k)
    factory = FailingEngine
res = train(args, algo, engine_factory=factory)
np.savez(out, params=res["params"], role=np.array(res["role"]), updates=res["updates"])
print("rank", rank, res["role"], "updates", res["updates"])
