"""The oracle against vectors produced by the REAL reference stack (jax / flax / optax / rlax), when they exist.

tests/golden/jax_vectors.npz is written by tools/make_jax_golden.py in an environment that has the reference's pinned libraries.  This image
has neither them nor a network, so the file is absent, these tests are skipped, and the oracle stays "parity unpinned" (DESIGN.md section 3).
Bars are the north-star ones: integer / index results bit-exact, floating point 1e-5."""
import os

import numpy as np
import pytest

from helpers import make_params

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jax_vectors.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/jax_vectors.npz not generated: run tools/make_jax_golden.py "
                                                                  "where jax 0.4.8 / flax 0.6.8 / optax 0.1.4 / rlax 0.1.5 are installed")
A = 18


@pytest.fixture(scope="module")
def G():
    return np.load(PATH)


def test_prng_streams(oracle, G):
    key = oracle.prng_key(1)
    assert np.array_equal(key, G["seed1_key"])
    assert np.array_equal(oracle.split(key, 4), G["seed1_split4"])
    assert np.array_equal(oracle.uniform(oracle.split(key, 4)[0], 16), G["seed1_uniform16"])
    assert np.array_equal(oracle.random_bits(oracle.prng_key(7), 9), G["bits_key7_n9"])
    for n in (5, 257, 1000, 15360):     # _shuffle round count, stable sort_key_val
        assert np.array_equal(oracle.permutation(oracle.prng_key(7), n), G[f"perm_key7_n{n}"]), n


def test_division_by_255_is_a_true_division(oracle, G):
    assert np.array_equal(np.array([oracle.u8_unit(x) for x in range(256)], np.float32), G["div255"])


def test_forward_and_sampling(oracle, G):
    P = make_params(A, int(G["fwd_params_seed"]))
    logits, value = oracle.nature_forward(P, A, G["fwd_obs"])
    np.testing.assert_allclose(logits, G["fwd_logits"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(value, G["fwd_value"], rtol=0, atol=1e-5)
    # sampling on the REFERENCE's logits: bit-exact actions need the same uniforms and a log that rounds like XLA's
    a, lp, key_out = oracle.sample_actions(G["fwd_logits"], oracle.prng_key(99))
    assert np.array_equal(key_out, G["sample_key_out"])
    assert np.array_equal(a, G["sample_actions"])
    np.testing.assert_allclose(lp, G["sample_logprob"], rtol=0, atol=1e-6)


def test_gae_and_advnorm(oracle, G):
    adv, tgt = oracle.gae(G["gae_r"], G["gae_v"], G["gae_d"], G["gae_nv"], G["gae_nd"])
    np.testing.assert_allclose(adv, G["gae_adv"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(tgt, G["gae_tgt"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(oracle.advnorm(adv, 4), G["gae_advnorm_jnp"], rtol=0, atol=1e-5)


def test_ppo_loss_head(oracle, G):
    st, dl, dv = oracle.ppo_loss_head(G["ppo_logits"], G["ppo_value"], G["ppo_actions"], G["ppo_oldlp"], G["ppo_adv"], G["ppo_tgt"])
    np.testing.assert_allclose(st, G["ppo_stats"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dl, G["ppo_dlogits"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(dv, G["ppo_dvalue"], rtol=0, atol=1e-6)


def test_vtrace_and_impala_loss_head(oracle, G):
    V = G["vt_V"]
    err, pg, q = oracle.vtrace(V[:-1], V[1:], G["vt_r"], G["vt_disc"], G["vt_rho"])
    np.testing.assert_allclose(err, G["vt_errors"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(pg, G["vt_pg"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(q, G["vt_q"], rtol=0, atol=1e-5)
    st, dl, dv = oracle.impala_loss_head(G["imp_logits"], G["imp_value"], G["imp_mu"], G["imp_actions"], G["imp_rewards"], G["imp_dones"],
                                         G["imp_first"])
    np.testing.assert_allclose(st, G["imp_stats"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dl, G["imp_dlogits"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(dv, G["imp_dvalue"], rtol=0, atol=1e-5)


def test_optimizers(oracle, G):
    import cleanba_amd.model as M
    n = G["adam_p0"].size
    p, m, v = G["adam_p0"].copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    for i, g in enumerate(G["adam_grads"]):       # inject_hyperparams evaluates the schedule at its pre-increment count
        lr = M.linear_schedule(i, 2.5e-4, int(G["adam_spu"]), int(G["adam_num_updates"]))
        assert abs(lr - G["adam_lrs"][i]) <= 1e-10
        oracle.adam_step(p, g, m, v, 0.5, lr, count=i + 1)
        np.testing.assert_allclose(p, G["adam_traj"][i], rtol=3e-7, atol=1e-7)
    q, nu = G["adam_p0"].copy(), np.zeros(n, np.float32)
    for i, g in enumerate(G["adam_grads"]):
        oracle.rmsprop_step(q, (30 * g).astype(np.float32), nu, 40.0, 6e-4)
        np.testing.assert_allclose(q, G["rms_traj"][i], rtol=3e-7, atol=2e-6)
    # MultiSteps(every_k=2): parameters move on every second call, by Adam on the running mean of the pair
    p, m, v = G["adam_p0"].copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    assert np.array_equal(G["multi_traj"][0], G["adam_p0"]) and np.array_equal(G["multi_traj"][2], G["multi_traj"][1])
    for k in range(2):
        g0, g1 = G["adam_grads"][2 * k].astype(np.float32), G["adam_grads"][2 * k + 1].astype(np.float32)
        acc = (g1 - g0) / np.float32(2) + g0                  # acc + (g - acc) / (mini_step + 1)
        oracle.adam_step(p, acc, m, v, 0.5, 2.5e-4, count=k + 1)
        np.testing.assert_allclose(p, G["multi_traj"][2 * k + 1], rtol=3e-7, atol=1e-7)


def test_max_pool_same_padding(G):
    import torch
    import torch.nn.functional as F
    pads = {84: (0, 1), 42: (0, 1), 21: (1, 1)}          # what oracle/cbm_oracle.c and tests/torch_ref.py assume for SAME, 3x3 / stride 2
    for hw, (lo, hi) in pads.items():
        x = torch.tensor(G[f"pool_in_{hw}"]).permute(0, 3, 1, 2)
        y = F.max_pool2d(F.pad(x, (lo, hi, lo, hi), value=float("-inf")), 3, 2).permute(0, 2, 3, 1).numpy()
        assert np.array_equal(y, G[f"pool_out_{hw}"]), hw
