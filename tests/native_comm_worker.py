"""One rank of tests/test_gpu_native_comm.py::test_native_allreduce_is_the_rank_ordered_sum: its own process and context on GPU 0 (ranks of one
process would share the runtime's few hardware queues, and a kernel that waits for a peer queued BEHIND it on the same queue can never see it).
Usage: python native_comm_worker.py <rank> <nranks> <port> <out.npz>"""
import os
import sys

import faulthandler

import numpy as np

faulthandler.dump_traceback_later(int(os.environ.get("CBM_WORKER_WATCHDOG_S", "180")), exit=True)   # a stalled rank says WHERE it stalled, then gets out of the way

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
rank, n, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]

import cleanba_amd.lib as L  # noqa: E402
from cleanba_amd import topology  # noqa: E402

# The rendezvous FIRST (it imports torch for the TCPStore), the HIP context after it — the trainer's order.  With the context first, both ranks of this test
# were found stalled inside `import torch` (torch/__init__.py loading its extension module with the HIP runtime already live in the process) on three of
# ten fresh boxes: the stack dumps of the watchdog above showed it (round 6).
rdv = topology.Rendezvous(n, rank, "127.0.0.1", port, timeout_s=120.0)
cfg = L.default_config(L.ALGO_PPO)
cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 8, 1, 8
ctx = L.Context(cfg)
blob = ctx.comm_native_export()
rdv.put(f"blob/{rank}", blob)
ctx.comm_native_init([blob if r == rank else bytes(rdv.get(f"blob/{r}")) for r in range(n)], rank)
assert ctx.comm_backend() == "native" and ctx.comm_size() == n


def grads_of(r, rep):
    rng = np.random.default_rng(1000 * rep + r)
    return (rng.normal(size=ctx.P) * 10.0 ** rng.integers(-6, 2, ctx.P)).astype(np.float32)


res = {}
for rep in range(3):                     # the flags only grow: repeated collectives on the same signal blocks
    ctx.write("grads", grads_of(rank, rep))
    ctx.sync()
    assert ctx.learner_allreduce_grads() == float(n)     # tail + head on the communication stream, learner stream joined
    ctx.sync()
    res[f"g{rep}"] = ctx.read("grads", np.float32)
vals = np.random.default_rng(77 + rank).normal(size=5)
for op in ("sum", "max", "min"):
    res[op] = ctx.comm_allreduce_f64(vals, op)
ctx.comm_barrier()
if os.environ.get("CBM_NATIVE_BENCH"):      # tools/native_allreduce_bench.py: ranks in lockstep, nothing else on the GPU
    import time
    for _ in range(5):
        ctx.learner_allreduce_grads()
    ctx.sync()
    ctx.comm_barrier()
    t0 = time.perf_counter()
    iters = 200
    for _ in range(iters):
        ctx.learner_allreduce_grads()
    ctx.sync()
    res["us_per_allreduce"] = np.float64((time.perf_counter() - t0) / iters * 1e6)
    res["bytes"] = np.int64(ctx.P * 4)
rdv.barrier("done")                      # nobody unmaps while a peer may still be inside a collective
np.savez(out, **res)
ctx.unmap_peers()                        # teardown order: unmap -> barrier -> free (include/cleanba_mi.h, EXPORT WINDOWS)
rdv.barrier("unmapped")
ctx.close()
print("rank", rank, "ok", flush=True)
