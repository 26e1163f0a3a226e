"""Gradient all-reduce overlapped with the backward pass (cleanba_amd.trainer.GradAllReducer, cbm_learner_stream_wait_tail /
cbm_learner_wait_stream) on ONE GPU.  RCCL needs two GPUs, so the collective is replaced by what SUM over two identical ranks does to
the buffer — multiply by two on the stream the collective would be ordered on — and the optimizer divides by 2 again (pmean, ppo:628).
x*2/2 is exact in fp32, so the run must equal the plain single-GPU update BIT FOR BIT; if the tail of the gradient were touched before
the dense weight gradient is final, or the optimizer ran before the communication stream is done, the bits differ.  Full-size
minibatches (3840 frames) keep the GPU inside the backward pass while the host issues the collective, 16 times per update."""
import numpy as np
import pytest

import cleanba_amd.lib as L
import cleanba_amd.model as M
import cleanba_amd.prng as prng

pytestmark = pytest.mark.gpu


class TwoIdenticalRanks:
    class ReduceOp:
        SUM = "sum"

    def __init__(self):
        self.calls = []

    def all_reduce(self, tensor, op=None, group=None):
        self.calls.append(tensor.numel())
        tensor.mul_(2.0)          # on torch's current stream, like the collective


@pytest.mark.parametrize("network,E,T", [("nature", 120, 128), ("impala_resnet", 16, 16)])
def test_overlapped_allreduce_equals_plain_update_bitwise(network, E, T):
    from cleanba_amd.trainer import GradAllReducer, HipEngine
    A = 18
    kind = L.NET_NATURE if network == "nature" else L.NET_IMPALA_RESNET

    def run(mode):
        cfg = L.default_config(L.ALGO_PPO)
        cfg.network, cfg.local_num_envs, cfg.num_actor_slots, cfg.num_steps = kind, E, 1, T
        cfg.actor_dense_ksplit = 14 if network == "nature" else 11
        eng = HipEngine(cfg)
        key = prng.prng_key(1)
        key, nk, ak, ck = prng.split(key, 4)
        eng.set_params(M.init_params(network, A, nk, ak, ck))
        eng.actor_set_key(0, key)
        eng.actor_env_reset_device(0, 3)
        out = []
        n_opt = 16
        bc = [M.adam_bias_corrections(i + 1) for i in range(2 * n_opt)]
        lkey = key.copy()
        fake = TwoIdenticalRanks()
        red = GradAllReducer(eng, 2, dist_module=fake, active=True, overlap=(mode == "overlap")) if mode != "plain" else None
        for u in range(2):
            eng.actor_begin_rollout(0, False)
            eng.actor_rollout_device(0, T)
            eng.actor_commit(0)
            eng.learner_wait()
            lrs = np.full(n_opt, 2.5e-4, np.float32)
            b1 = np.array([b[0] for b in bc[u * n_opt:(u + 1) * n_opt]], np.float32)
            b2 = np.array([b[1] for b in bc[u * n_opt:(u + 1) * n_opt]], np.float32)
            if mode == "plain":
                lkey, _ = eng.learner_update(lkey, lrs, b1, b2, False)
            else:
                lkey = eng.learner_prepare(lkey)
                i = 0
                for e in range(4):
                    lkey = eng.learner_epoch_begin(lkey)
                    for mb in range(4):
                        eng.learner_minibatch_grad(e, mb)
                        div = red()
                        eng.learner_optimizer_step(float(lrs[i]), float(b1[i]), float(b2[i]), div)
                        i += 1
                eng.learner_finish(n_opt, False)
            out.append(eng.get_params())
        calls = fake.calls
        eng.close()
        return out, calls

    plain, _ = run("plain")
    flat, calls_flat = run("flat")
    over, calls_over = run("overlap")
    P = plain[0].size
    assert calls_flat[:2] == [P, P]
    assert calls_over[0] + calls_over[1] == P and calls_over[0] > 0.9 * P      # tail first (dense + heads), then the small head
    for u in range(2):
        assert np.isfinite(plain[u]).all()
        assert np.array_equal(plain[u], flat[u]), "split-form update with a flat all-reduce differs from cbm_learner_update"
        assert np.array_equal(plain[u], over[u]), "overlapped all-reduce changed the result: a stream-ordering bug"
    assert not np.array_equal(plain[0], plain[1])
