"""Pins the oracle's PRNG restatement and the shared scalar math."""
import math

import numpy as np


def test_threefry_random123_kat(oracle):
    o = oracle
    assert list(o.threefry2x32([0, 0], [0, 0])) == [0x6B200159, 0x99BA4EFE]
    assert list(o.threefry2x32([0xFFFFFFFF] * 2, [0xFFFFFFFF] * 2)) == [0x1CB996FC, 0xBB002BE7]
    assert list(o.threefry2x32([0x13198A2E, 0x03707344], [0x243F6A88, 0x85A308D3])) == [0xC4923A9C, 0x483DF7A0]


def test_jax_documented_split(oracle):
    # jax docs: split(PRNGKey(0)) == [[4146024105, 967050713], [2718843009, 1272950319]]
    k = oracle.prng_key(0)
    assert list(k) == [0, 0]
    s = oracle.split(k, 2)
    assert s.tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]


def test_regression_seed1_keys(oracle):
    # KAT-3: key, network_key, actor_key, critic_key at --seed 1 (ppo:468-469)
    s = oracle.split(oracle.prng_key(1), 4)
    assert s.tolist() == [[869452973, 4133157646], [261504626, 4112007671], [3597360905, 253918841],
                          [98387565, 678776088]]
    u = oracle.uniform(oracle.prng_key(0), 3)
    np.testing.assert_allclose(u, [0.9653214, 0.31468165, 0.63302994], rtol=0, atol=1e-7)


def test_uniform_range_and_odd_padding(oracle):
    for n in (1, 2, 5, 120 * 18, 120 * 18 + 1):
        u = oracle.uniform(oracle.prng_key(7), n)
        assert u.shape == (n,) and (u >= 0).all() and (u < 1).all()
    # odd n pads one zero counter and drops the last output: first half unaffected by the pad
    a = oracle.random_bits(oracle.prng_key(3), 5)
    assert len(set(a.tolist())) == 5


def test_permutation_is_stable_two_round_sort(oracle):
    n = 15360
    assert oracle.lib().cbo_shuffle_rounds(n) == 2 and oracle.lib().cbo_shuffle_rounds(30720) == 2
    key = oracle.prng_key(11)
    perm = oracle.permutation(key, n)
    assert sorted(perm.tolist()) == list(range(n))
    # numpy restatement: two rounds of stable argsort with split keys
    x = np.arange(n)
    k = key
    for _ in range(2):
        ks = oracle.split(k, 2)
        k, sub = ks[0], ks[1]
        bits = oracle.random_bits(sub, n)
        x = x[np.argsort(bits, kind="stable")]
    assert (x == perm).all()


def test_u8_unit_exact_division(oracle):
    for x in range(256):
        assert np.float32(oracle.u8_unit(x)) == np.float32(x) / np.float32(255.0)


def test_logf_expf_accuracy(oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(1e-7, 20, 2000), 2.0 ** rng.uniform(-126, 126, 2000), [1.0, 0.5, 2.0]]).astype(np.float32)
    for x in xs:
        got, ref = oracle.logf(float(x)), math.log(float(x))
        assert abs(got - ref) <= 2.5e-7 * max(1.0, abs(ref)) + 1e-7 * abs(ref), (x, got, ref)
    assert oracle.logf(0.0) == -math.inf and oracle.logf(math.inf) == math.inf and math.isnan(oracle.logf(-1.0))
    for x in np.concatenate([rng.uniform(-87, 88, 3000), [0.0, -100.0, -87.4]]).astype(np.float32):
        got = oracle.expf(float(x))
        ref = math.exp(float(x)) if x > -87.3 else 0.0
        assert abs(got - ref) <= 3e-7 * ref + 1e-45, (x, got, ref)
    assert oracle.expf(100.0) == math.inf
