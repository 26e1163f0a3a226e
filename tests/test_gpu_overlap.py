"""Gradient all-reduce overlapped with the backward pass (cbm_learner_allreduce_grads inside cbm_learner_update, csrc/comm.hip) on ONE GPU.

(1) Stream ordering.  RCCL between ranks needs two GPUs, so the self-test communicator (cbm_comm_init_loopback) stands in: it does to the
    buffer what SUM over n identical ranks does — multiply by n, on the same communication stream and behind the same events as the RCCL
    calls — and the optimizer divides by n again (pmean, ppo:628).  x*2/2 is exact in fp32, so the run must equal the plain single-GPU
    update BIT FOR BIT; if the tail of the gradient were touched before the dense weight gradient is final, or the optimizer ran before
    the communication stream is done, the bits differ.  Full-size minibatches (3840 frames) keep the GPU inside the backward pass while
    the host issues the collective, 16 times per update.
(2) The real thing at world size 1: a one-rank RCCL communicator (CBM_FORCE_DIST=1 in the trainer / bench) must also reproduce the plain
    update bit for bit, through librccl's own kernels and launch path."""
import numpy as np
import pytest

import cleanba_amd.lib as L
import cleanba_amd.model as M
import cleanba_amd.prng as prng

pytestmark = pytest.mark.gpu


def _run(network, E, T, mode, updates=2):
    from cleanba_amd.trainer import HipEngine
    A = 18
    kind = L.NET_NATURE if network == "nature" else L.NET_IMPALA_RESNET
    cfg = L.default_config(L.ALGO_PPO)
    cfg.network, cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = kind, E, 1, T
    cfg.actor_dense_ksplit = 14 if network == "nature" else 11
    eng = HipEngine(cfg)
    if mode == "loopback2":
        eng.comm_init_loopback(2)
    elif mode == "rccl1":
        eng.comm_init(eng.comm_unique_id(), 1, 0)
    assert eng.comm_size() == {"plain": 0, "loopback2": 2, "rccl1": 1}[mode]
    key = prng.prng_key(1)
    key, nk, ak, ck = prng.split(key, 4)
    eng.set_params(M.init_params(network, A, nk, ak, ck))
    eng.actor_set_key(0, key)
    eng.actor_env_reset_device(0, 3)
    out, stats_out = [], []
    n_opt = 16
    bc = [M.adam_bias_corrections(i + 1) for i in range(updates * n_opt)]
    lkey = key.copy()
    for u in range(updates):
        eng.actor_begin_rollout(0, False)
        eng.actor_rollout_device(0, T)
        eng.actor_commit(0)
        eng.learner_wait()
        lrs = np.full(n_opt, 2.5e-4, np.float32)
        b1 = np.array([b[0] for b in bc[u * n_opt:(u + 1) * n_opt]], np.float32)
        b2 = np.array([b[1] for b in bc[u * n_opt:(u + 1) * n_opt]], np.float32)
        lkey, stats = eng.learner_update(lkey, lrs, b1, b2, True)
        out.append(eng.get_params())
        stats_out.append(stats)
    eng.close()
    return out, stats_out


@pytest.mark.parametrize("network,E,T", [("nature", 120, 128), ("impala_resnet", 16, 16)])
def test_overlapped_allreduce_equals_plain_update_bitwise(network, E, T, monkeypatch):
    plain, pstats = _run(network, E, T, "plain")
    over, ostats = _run(network, E, T, "loopback2")
    monkeypatch.setenv("CBM_ALLREDUCE_OVERLAP", "0")
    flat, _ = _run(network, E, T, "loopback2")
    for u in range(2):
        assert np.isfinite(plain[u]).all()
        assert np.array_equal(plain[u], flat[u]), "update with one flat all-reduce after the backward pass differs from the plain update"
        assert np.array_equal(plain[u], over[u]), "overlapped all-reduce changed the result: a stream-ordering bug"
        assert np.array_equal(pstats[u], ostats[u])       # pmean of identical ranks' statistics = the statistics
    assert not np.array_equal(plain[0], plain[1])


def test_one_rank_rccl_communicator_equals_plain_update_bitwise():
    plain, pstats = _run("nature", 120, 128, "plain")
    rccl, rstats = _run("nature", 120, 128, "rccl1")
    for u in range(2):
        assert np.array_equal(plain[u], rccl[u])
        assert np.array_equal(pstats[u], rstats[u])


def test_comm_host_values_and_barrier():
    from cleanba_amd.trainer import HipEngine
    cfg = L.default_config(L.ALGO_PPO)
    cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = 8, 1, 8
    eng = HipEngine(cfg)
    with pytest.raises(L.CbmError, match="not initialised"):
        eng.comm_barrier()
    eng.comm_init(eng.comm_unique_id(), 1, 0)
    eng.comm_barrier()
    assert np.array_equal(eng.comm_allreduce_f64([1.5, -2.0], "max"), [1.5, -2.0])
    assert np.array_equal(eng.comm_allreduce_f64([1.5, -2.0], "sum"), [1.5, -2.0])
    eng.close()
