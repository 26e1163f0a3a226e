"""Split actor/learner path on ONE GPU: an actor-only context and a learner-only context (two threads of one process, both on
cuda:0) exchange rollout shards and parameters through a loopback stand-in for torch.distributed.  This exercises everything of the
split path that lives in the library and in HipEngine — cbm_ingest_begin/commit, cbm_params_publish_external, the zero-copy ring
views, fences and io streams — and must reproduce the ordinary single-process run bit for bit (with one learner nothing is
re-sharded).  RCCL itself needs two GPUs; the same code runs over gloo with world sizes 2-4 in tests/test_host_cpu.py."""
import copy
import os
import queue
import threading
from collections import defaultdict

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Shared:
    def __init__(self):
        self.q = defaultdict(queue.Queue)
        self.lock = threading.Lock()


class LoopbackDist:
    """send/recv between threads; tensors are cloned on the sender's current stream and copied on the receiver's."""

    class ReduceOp:
        SUM = "sum"

    def __init__(self, shared, rank):
        self.sh, self.rank, self.n_groups = shared, rank, 0

    def new_group(self, ranks):
        self.n_groups += 1
        return ("group", self.n_groups, tuple(ranks))

    def send(self, tensor, dst, group=None):
        import torch
        t = tensor.clone()
        torch.cuda.current_stream().synchronize()
        self.sh.q[(self.rank, dst, group)].put(t)

    def recv(self, tensor, src, group=None):
        import torch
        t = self.sh.q[(src, self.rank, group)].get(timeout=300)
        tensor.copy_(t)
        torch.cuda.current_stream().synchronize()   # `t` belongs to the sender's stream in torch's caching allocator: finish before dropping it


@pytest.mark.parametrize("algo", ["ppo", "impala"])
def test_split_loopback_equals_single_process(tmp_path, algo):
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import train
    os.chdir(str(tmp_path))
    E, T, updates = 8, 8, 3
    base = ["--local-num-envs", str(E), "--num-actor-threads", "2", "--num-steps", str(T), "--env-backend", "device", "--network", "nature",
            "--total-timesteps", str(updates * E * 2 * T), "--log-frequency", "1000", "--update-epochs", "1"]
    ref = train(parse_args(base, algo), algo)

    split_argv = base + ["--distributed", "--actor-device-ids", "0", "--learner-device-ids", "1"]
    shared, results, errors = _Shared(), {}, []

    def run(rank):
        try:
            args = parse_args(split_argv, algo)
            results[rank] = train(copy.deepcopy(args), algo, rendezvous=(2, rank, 0, None, None), dist_module=LoopbackDist(shared, rank))
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            raise

    ths = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(2)]
    [t.start() for t in ths]
    [t.join(timeout=600) for t in ths]
    assert not errors, errors
    assert results[0]["role"] == "actor" and results[1]["role"] == "learner0"
    assert results[1]["updates"] == updates == ref["updates"]
    assert np.array_equal(results[1]["params"], ref["params"])
    assert np.array_equal(results[0]["params"], results[1]["params"])
