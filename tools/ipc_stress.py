"""Stress of the HIP-IPC control plane on ONE GPU: N processes, each cycling K times through what a `bench.py --gpus N` rank does between its
two phases — create a context, export the ring fields / parameter buffers / native-communicator blob, map every peer's, run a collective and a
peer copy, tear down — in the SAME processes, so that whatever a previous cycle leaves behind in the runtime (its IPC socket server, mappings
closed after the owner freed, address reuse) meets the next cycle's fresh exports.  One line per failure with rank / cycle / step, a
summary line per rank.

usage:  python tools/ipc_stress.py <nprocs> <cycles> [mode]        (parent; spawns the ranks)
  mode: "racy"   — every rank tears down as soon as it is done (no barrier between 'peers unmapped' and 'owner frees': the round-4 shape)
        "safe"   — unmap -> barrier -> free (what the product does now)
        "noclose"— never unmap peers' buffers before the owner frees them (mappings die with the context)"""
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def worker(rank, n, port, cycles, mode):
    import cleanba_amd.lib as L
    from cleanba_amd import topology
    from cleanba_amd.trainer import HipEngine
    fields = topology.PPO_FIELDS
    fails = 0
    t0 = time.time()
    for cyc in range(cycles):
        rdv = topology.Rendezvous(n, rank, "127.0.0.1", port, timeout_s=120.0, prefix=f"stress{cyc}")
        step = "create"
        try:
            cfg = L.default_config(L.ALGO_PPO)
            cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 8, 1, 8
            cfg.device = int(os.environ.get("CBM_FORCE_DEVICE", 0))
            ctx = HipEngine(cfg)
            step = "export"
            import pickle
            rdv.put(f"ring/{rank}", pickle.dumps(ctx.export_ring(fields)))
            rdv.put(f"ap/{rank}", pickle.dumps(ctx.export_actor_params()))
            blob = ctx.comm_native_export()
            rdv.put(f"blob/{rank}", blob)
            step = "open"
            rings, aps = {}, {}
            for r in range(n):
                if r == rank:
                    continue
                step = f"open ring of {r}"
                rings[r] = ctx.open_peer_ring(pickle.loads(rdv.get(f"ring/{r}")))
                step = f"open params of {r}"
                aps[r] = ctx.open_peer_params(pickle.loads(rdv.get(f"ap/{r}")))
            step = "native init"
            ctx.comm_native_init([blob if r == rank else bytes(rdv.get(f"blob/{r}")) for r in range(n)], rank)
            step = "collective"
            g = np.full(ctx.P, float(rank + 1), np.float32)
            ctx.write("grads", g)
            ctx.sync()
            ctx.learner_allreduce_grads()
            ctx.sync()
            got = ctx.read("grads", np.float32)
            want = float(n * (n + 1) // 2)
            if not np.all(got == want):
                raise RuntimeError(f"all-reduce result {got[:4]} != {want}")
            ctx.comm_barrier()
            step = "teardown"
            if mode == "safe":
                ctx.unmap_peers()
                rdv.barrier("unmapped")
            elif mode == "racy":
                time.sleep(0.002 * ((rank * 7 + cyc) % 5))   # spread the teardown order
            ctx.close()
        except BaseException as e:  # noqa: BLE001
            fails += 1
            print(f"FAIL rank {rank} pid {os.getpid()} cycle {cyc} step '{step}': {type(e).__name__}: {e}", flush=True)
            try:
                rdv.abort(f"cycle {cyc} step {step}")
            except Exception:  # noqa: BLE001
                pass
            break
    print(f"rank {rank} pid {os.getpid()}: {cycles if not fails else cyc} cycles ok, {fails} failure(s), {time.time() - t0:.1f} s, mode {mode}", flush=True)
    os._exit(1 if fails else 0)


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
        return
    n, cycles = int(sys.argv[1]), int(sys.argv[2])
    mode = sys.argv[3] if len(sys.argv) > 3 else "safe"
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CBM_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(n), str(port), str(cycles), mode], env=env) for r in range(n)]
    codes = [p.wait() for p in procs]
    print(f"ipc_stress n={n} cycles={cycles} mode={mode}: exit codes {codes}", flush=True)
    sys.exit(max(abs(c) for c in codes))


if __name__ == "__main__":
    main()
