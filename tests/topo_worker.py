"""One ROLE process of a split-topology run, on the HIP engine (every role pinned to GPU 0: CBM_FORCE_DEVICE, learner all-reduce = the
library's native backend) or on the CPU oracle engine (gloo) — same host program, same arguments, so the two can be compared.
Usage: python topo_worker.py <rank> <world> <port> <out.npz> <algo> <hip|oracle> <E> <T> <updates> <actor_ids:learner_ids> <update_epochs> [env_id]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

rank, world, port, out, algo, engine = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
E, T, updates = (int(x) for x in sys.argv[7:10])
aids, lids = (x.split(",") for x in sys.argv[10].split(":"))
epochs = sys.argv[11]
env_id = sys.argv[12] if len(sys.argv) > 12 else "Breakout-v5"
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
if engine == "hip":
    os.environ.update(CBM_FORCE_DEVICE="0", LOCAL_RANK="0")

from cleanba_amd.args import parse_args  # noqa: E402
from cleanba_amd.trainer import train  # noqa: E402

# the device env and the host env are byte-identical twins and actions are bit-exact, so both engines see the same first rollout
argv = ["--local-num-envs", str(E), "--num-actor-threads", "1", "--num-steps", str(T), "--env-backend", "device" if engine == "hip" else "host",
        "--network", "nature", "--env-id", env_id, "--total-timesteps", str(updates * E * T * (world // (len(aids) + len(lids)))), "--log-frequency", "1",
        "--update-epochs", epochs, "--distributed",
        "--actor-device-ids"] + aids + ["--learner-device-ids"] + lids
os.chdir(os.environ.get("CBM_TEST_TMP", "/tmp"))
factory = None
if engine == "oracle":
    from oracle_engine import OracleEngine
    factory = OracleEngine
seen = {}


def on_update(v, stats, eng):
    if v == 1:
        if hasattr(eng, "comm_backend") and eng.comm_size() > 0:
            print("allreduce.backend:", eng.comm_backend(), "ranks", eng.comm_size(), flush=True)
        seen["stats"] = np.asarray(stats, np.float32).copy()


res = train(parse_args(argv, algo), algo, engine_factory=factory, on_update=on_update)
import cleanba_amd.model as M  # noqa: E402
import cleanba_amd.prng as prng  # noqa: E402
key = prng.prng_key(1)
_, nk, ak, ck = prng.split(key, 4)
np.savez(out, params=res["params"], role=np.array(res["role"]), updates=res["updates"], stats=seen.get("stats", np.zeros(0, np.float32)),
         p0=M.init_nature_params(18, nk, ak, ck))
print("rank", rank, res["role"], "updates", res["updates"], flush=True)
