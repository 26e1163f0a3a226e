"""The library's own all-reduce (csrc/comm.hip, "native": kernels over IPC-mapped / same-process peer buffers, no RCCL) — the backend that lets
several learner ranks share ONE GPU, which RCCL refuses.  Replaces jax.lax.pmean over the learner devices (ppo:628,649-653).

(1) n rank processes on GPU 0: DIFFERENT gradients per rank, result == ((g0 + g1) + g2) bit for bit on every rank (the fixed rank order is
    the determinism ppo:30 asks of XLA), for the overlapped tail / head pair and for the f64 sum / max / min used by barriers and timing;
(2) a dead peer: the flag wait times out and the next synchronising call returns an error instead of hanging the GPU;
(3) BASELINE configs[3] for real on one GPU: `a0-l1,2,3` as four ROLE PROCESSES on GPU 0 at E = 120, T = 128 — every learner holds a different
    40-env shard, the gradients that are summed differ — against the same topology on the CPU oracle engine over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cleanba_amd.lib as L

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("n", [2, 3, 5])
def test_native_allreduce_is_the_rank_ordered_sum(tmp_path, n):
    """n rank processes on GPU 0 with DIFFERENT data: after the overlapped tail / head pair every rank holds ((g0 + g1) + g2) ... bit for bit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CBM_NATIVE_TIMEOUT_S="60")
    outs = [os.path.join(str(tmp_path), f"r{r}.npz") for r in range(n)]

    def attempt():
        port = _free_port()
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "native_comm_worker.py"), str(r), str(n), str(port), outs[r]], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(n)]
        logs, ok = [], True
        for p in procs:
            try:
                logs.append(p.communicate(timeout=300)[0].decode())
            except subprocess.TimeoutExpired:
                p.kill()
                logs.append("TIMED OUT after 300 s\n" + (p.communicate()[0] or b"").decode())
            ok = ok and p.returncode == 0
        return ok, logs

    # Round 6: on three of ten fresh boxes the two rank processes of this test did not finish.  The workers' watchdog (stack dump after 180 s) showed both inside
    # `import torch` — topology.Rendezvous importing torch for its TCPStore AFTER the worker had created its HIP context; the worker (and bench.py's run_dp) now
    # build the rendezvous first, the trainer's order.  The retry stays as a net: a cold-start stall must not stand in front of the suite's parity result (pytest -x).
    ok, logs = attempt()
    if not ok:
        print("FIRST ATTEMPT of the native all-reduce rank processes failed; retrying once.  Logs of the first attempt:\n" + "\n=====\n".join(lg[-3000:] for lg in logs))
        ok, logs = attempt()
    assert ok, "\n=====\n".join(lg[-2000:] for lg in logs)
    got = [np.load(o) for o in outs]
    P = got[0]["g0"].size
    for rep in range(3):
        want = None
        for r in range(n):
            rng = np.random.default_rng(1000 * rep + r)
            g = (rng.normal(size=P) * 10.0 ** rng.integers(-6, 2, P)).astype(np.float32)
            want = g if want is None else want + g
        for r in range(n):
            assert np.array_equal(got[r][f"g{rep}"].view(np.uint32), want.view(np.uint32)), (rep, r, np.abs(got[r][f"g{rep}"] - want).max())
    vals = [np.random.default_rng(77 + r).normal(size=5) for r in range(n)]
    wsum = vals[0].copy()
    for r in range(1, n):
        wsum = wsum + vals[r]
    for r in range(n):
        assert np.array_equal(got[r]["sum"], wsum)
        assert np.array_equal(got[r]["max"], np.max(np.stack(vals), axis=0)) and np.array_equal(got[r]["min"], np.min(np.stack(vals), axis=0))


def test_native_allreduce_dead_peer_times_out_with_an_error():
    """Run in a child process: CBM_NATIVE_TIMEOUT_S is read per launch, and a timed-out communicator is not reusable."""
    code = r'''
import os, sys, time
sys.path.insert(0, %r)
os.environ["CBM_NATIVE_TIMEOUT_S"] = "1.5"
import numpy as np
import cleanba_amd.lib as L
ctxs = []
for _ in range(2):
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 8, 1, 8
    ctxs.append(L.Context(cfg))
blobs = [c.comm_native_export() for c in ctxs]
for r, c in enumerate(ctxs):
    c.comm_native_init(blobs, r)
t0 = time.time()
ctxs[0].learner_allreduce_grads()        # rank 1 never joins
try:
    ctxs[0].sync()
except RuntimeError as e:
    assert "timed out" in str(e), e
    assert 1.0 < time.time() - t0 < 30.0, time.time() - t0
    print("TIMEOUT-REPORTED", flush=True)
else:
    raise SystemExit("a collective without its peer completed")
os._exit(0)
''' % os.path.dirname(HERE)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0 and b"TIMEOUT-REPORTED" in p.stdout, p.stdout.decode()[-3000:]


def _run_topology(engine, tmp, tag, E, T, updates, ids, epochs, omp=None, groups=1, env_id="Breakout-v5", algo="ppo"):
    aids, lids = ids.split(":")
    world = groups * (len(aids.split(",")) + len(lids.split(",")))
    port = _free_port()
    env = dict(os.environ, CBM_TEST_TMP=str(tmp), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if omp:
        env["OMP_NUM_THREADS"] = str(omp)
    outs, procs = [], []
    for r in range(world):
        out = os.path.join(str(tmp), f"{tag}_{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "topo_worker.py"), str(r), str(world), str(port), out, algo, engine, str(E), str(T),
                                       str(updates), ids, str(epochs), env_id], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=1500)[0].decode() for p in procs]
    for p in procs:
        assert p.returncode == 0, "\n=====\n".join(lg[-2500:] for lg in logs)
    return [np.load(o) for o in outs], logs


def test_a0_l123_small_native_equals_oracle_topology(tmp_path):
    """The same four role processes at a size that runs in seconds (E = 12, T = 8, two updates): learners agree bit for bit with each other, and
    with the CPU oracle topology within the backward tolerance after the FIRST update (the second one starts from different bits)."""
    (a, l0, l1, l2), logs = _run_topology("hip", tmp_path, "hip", 12, 8, 1, "0:1,2,3", 2)
    assert "allreduce.backend: native" in "".join(logs)
    assert np.array_equal(l0["params"], l1["params"]) and np.array_equal(l0["params"], l2["params"])
    assert np.array_equal(a["params"], l0["params"])
    (_, o0, _, _), _ = _run_topology("oracle", tmp_path, "cpu", 12, 8, 1, "0:1,2,3", 2, omp=4)
    d = np.abs(l0["params"] - o0["params"])
    print("a0-l1,2,3 small: |p - p_oracle| max %.2e" % d.max())
    assert np.abs(o0["params"] - o0["p0"]).max() > 1e-4 and d.max() <= 1e-5
    np.testing.assert_allclose(l0["stats"], o0["stats"], rtol=1e-5, atol=1e-6)


def test_a0_l12_impala_native_equals_oracle_topology(tmp_path):
    """The IMPALA script in a split topology (impala:599-645: V-trace loss, RMSProp, pmean of four statistics): one actor + two learner role processes
    on GPU 0, each learner a different half of the env columns, gradients and statistics through the native all-reduce; after the first update against
    the same topology on the CPU oracle engine over gloo."""
    (a, l0, l1), logs = _run_topology("hip", tmp_path, "ihip", 8, 8, 1, "0:1,2", 1, algo="impala")
    assert "allreduce.backend: native ranks 2" in "".join(logs)
    assert np.array_equal(l0["params"], l1["params"]) and np.array_equal(a["params"], l0["params"])
    (_, o0, _), _ = _run_topology("oracle", tmp_path, "icpu", 8, 8, 1, "0:1,2", 1, omp=4, algo="impala")
    d = np.abs(l0["params"] - o0["params"])
    print("impala a0-l1,2 small: |p - p_oracle| max %.2e" % d.max())
    assert np.abs(o0["params"] - o0["p0"]).max() > 1e-5 and d.max() <= 1e-5
    np.testing.assert_allclose(l0["stats"], o0["stats"], rtol=1e-5, atol=1e-5)   # (IMPALA's losses are SUMS over T x B, impala:569-597: the bar of tests/test_gpu_parity.py)


def test_configs4_two_groups_atari57_mix_native_equals_oracle_topology(tmp_path):
    """BASELINE configs[4] (benchmark.sh:80 family: `2x(a0-l1,2,3)`, `--distributed` with a split layout, synthetic Atari-57 frame mix) as EIGHT role
    processes on GPU 0: two actors with different env seeds and game sets, six learners in ONE communicator (gradients averaged over the learners of
    BOTH groups through the native all-reduce), each group's learner 0 feeding its own actor.  Small sizes; against the same topology on the CPU oracle
    engine over gloo after the first update."""
    outs, logs = _run_topology("hip", tmp_path, "hip8", 12, 8, 1, "0:1,2,3", 2, groups=2, env_id="Atari57Mix-v5")
    a0, l00, l01, l02, a1, l10, l11, l12 = outs
    assert "allreduce.backend: native ranks 6" in "".join(logs)
    for x in (l01, l02, l10, l11, l12):
        assert np.array_equal(l00["params"], x["params"])
    assert np.array_equal(a0["params"], l00["params"]) and np.array_equal(a1["params"], l10["params"])
    o, _ = _run_topology("oracle", tmp_path, "cpu8", 12, 8, 1, "0:1,2,3", 2, omp=2, groups=2, env_id="Atari57Mix-v5")
    d = np.abs(l00["params"] - o[1]["params"])
    print("2x(a0-l1,2,3) small, Atari-57 mix: |p - p_oracle| max %.2e" % d.max())
    assert np.abs(o[1]["params"] - o[1]["p0"]).max() > 1e-4 and d.max() <= 1e-5
    np.testing.assert_allclose(l00["stats"], o[1]["stats"], rtol=1e-5, atol=1e-6)


@pytest.mark.slow
def test_configs3_a0_l123_full_size_on_one_gpu(tmp_path):
    """BASELINE configs[3] (README.md:62: `--actor-device-ids 0 --learner-device-ids 1 2 3`) at E = 120, T = 128 as four role processes on GPU 0:
    B_dev = 40 per learner, 1280-frame minibatches, a REAL reduction of three different gradients per minibatch (the native all-reduce), shards
    written by IPC peer copies.  One whole update against the same topology on the CPU oracle engine (gloo all-reduce) at the whole-update bars."""
    (a, l0, l1, l2), logs = _run_topology("hip", tmp_path, "hip", 120, 128, 1, "0:1,2,3", 4)
    assert "allreduce.backend: native" in "".join(logs)
    assert np.array_equal(l0["params"], l1["params"]) and np.array_equal(l0["params"], l2["params"])
    assert np.array_equal(a["params"], l0["params"])
    (_, o0, o1, _), _ = _run_topology("oracle", tmp_path, "cpu", 120, 128, 1, "0:1,2,3", 4, omp=max(4, (os.cpu_count() or 16) // 4))
    assert np.array_equal(o0["params"], o1["params"])
    d = np.abs(l0["params"] - o0["params"])
    serr = np.abs(l0["stats"] - o0["stats"]) / np.maximum(np.abs(o0["stats"]), 1e-3)
    print("configs[3] on one GPU: |p - p_oracle| median %.2e, 99.99 %% quantile %.2e, max %.2e; statistics worst relative error per column %s"
          % (np.median(d), np.quantile(d, 0.9999), d.max(), serr.max(axis=0)))
    assert np.abs(o0["params"] - o0["p0"]).max() > 1e-4
    # statistics: north_star's 1e-5 relative; absolute 2.5e-7 for the columns that are ~1e-3 themselves (policy loss, approx_kl — the latter a diagnostic, the
    # mean of (ratio - 1) - log ratio: measured worst 1.4e-7 absolute on one of its 16 rows, 5.8e-5 of its 2.4e-3)
    np.testing.assert_allclose(l0["stats"], o0["stats"], rtol=1e-5, atol=2.5e-7)
    # the role processes run the DEFAULT configuration: 1280-frame minibatches take the exact-product conv1 kernels (cbm_config.conv1_fp32_chain = 0), whose
    # forward rounds in another order than the oracle's chain — the parameter bars are those of the whole-update test in that mode
    # (tests/test_gpu_fullsize_oracle.py, chain = 0: ReLU flips of pre-activations within ~1e-7 of zero through Adam's 1 / (sqrt(v) + eps));
    # (the oracle role processes restate that conv1 with the measured rule of the bf16 matrix instruction: tests/oracle_engine.py);
    # measured here: median 0, 99.99 % quantile 9.5e-7, max 5.7e-6 (against the chain oracle: 1.6e-6 / 1.03e-5)
    assert np.median(d) <= 1e-7 and np.quantile(d, 0.9999) <= 2.5e-5 and d.max() <= 5e-5, (np.median(d), np.quantile(d, 0.9999), d.max())


def test_export_windows_survive_repeated_create_map_free_cycles_in_the_same_processes():
    """Round 4's red driver run: `bench.py --gpus 8` freed the gradient buffers of its data-parallel phase while peers still mapped them, and the SAME
    processes' next exports (the topology phase) could not be mapped ("hipIpcOpenMemHandle: invalid device pointer").  tools/ipc_stress.py cycles
    four processes six times through create -> export windows -> map every peer -> native all-reduce -> unmap -> barrier -> free (the order the product
    keeps now); with the round-4 order ('racy') the second cycle fails on this stack."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "ipc_stress.py"), "4", "6", "safe"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0 and "FAIL" not in out, out[-3000:]
    assert out.count("6 cycles ok, 0 failure(s)") == 4, out[-3000:]


def test_native_communicator_cannot_be_initialised_again_after_close():
    """cbm_ipc_close_all leaves the slot's signal block (old sequence numbers, sticky error words) in place, and a re-init restarted the
    sequence at 0 under flags that only grow — now the slot refuses; a one-rank communicator is enough to show it."""
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 8, 1, 8
    ctx = L.Context(cfg)
    try:
        blob = ctx.comm_native_export()
        ctx.comm_native_init([blob], 0)
        assert ctx.comm_backend() == "native"
        ctx.comm_barrier()
        ctx.unmap_peers()
        with pytest.raises(L.CbmError, match="closed"):
            ctx.comm_native_init([blob], 0)
    finally:
        ctx.close()
