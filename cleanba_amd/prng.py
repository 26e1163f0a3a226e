"""jax.random-compatible key arithmetic for the host side (numpy, vectorised).

The reference seeds everything from `jax.random.PRNGKey(args.seed)` and `jax.random.split`
(cleanba_ppo.py:468-470, 677): this module reproduces those uint32 key streams so that the
actor / learner keys handed to the HIP library are the ones the reference would use.
(threefry2x32, 20 rounds; jax 0.4.8's non-partitionable `split` / `random_bits` layout.)
"""
import numpy as np

_R0 = (13, 15, 26, 6)
_R1 = (17, 29, 16, 24)
_M = np.uint64(0xFFFFFFFF)


def _rotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M


def threefry2x32(k0, k1, c0, c1):
    """All arguments uint32 scalars or arrays; returns (o0, o1) uint32 arrays."""
    k0, k1 = np.uint64(k0), np.uint64(k1)
    x0 = (np.asarray(c0, np.uint64) + k0) & _M
    x1 = (np.asarray(c1, np.uint64) + k1) & _M
    ks = (k0, k1, k0 ^ k1 ^ np.uint64(0x1BD11BDA))
    for i in range(5):
        for r in (_R0 if i % 2 == 0 else _R1):
            x0 = (x0 + x1) & _M
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(i + 1) % 3]) & _M
        x1 = (x1 + ks[(i + 2) % 3] + np.uint64(i + 1)) & _M
    return x0.astype(np.uint32), x1.astype(np.uint32)


def prng_key(seed):
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], np.uint32)


def random_bits(key, n):
    half = (n + 1) // 2
    c0 = np.arange(half, dtype=np.uint64)
    c1 = c0 + np.uint64(half)
    c1[c1 >= n] = 0
    o0, o1 = threefry2x32(key[0], key[1], c0, c1)
    return np.concatenate([o0, o1])[:n]


def split(key, num=2):
    return random_bits(key, 2 * num).reshape(num, 2)


def uniform(key, n):
    bits = random_bits(key, n)
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    return np.maximum(f, np.float32(0.0))
