"""Shared test helpers: deterministic synthetic params / frames (pure numpy, seeded)."""
import numpy as np


def make_params(A, seed=0, scale=1.0):
    """Random parameters at realistic magnitudes (orthogonal-like std), NOT the product init."""
    rng = np.random.default_rng(seed)
    shapes = [((8, 8, 4, 32), 256), ((32,), 0), ((4, 4, 32, 64), 512), ((64,), 0), ((3, 3, 64, 64), 576), ((64,), 0),
              ((3136, 512), 3136), ((512,), 0), ((512, A), 512), ((A,), 0), ((512, 1), 512), ((1,), 0)]
    parts = []
    for shp, fan in shapes:
        if fan:
            parts.append(rng.normal(0, scale * np.sqrt(2.0 / fan), size=shp).astype(np.float32).ravel())
        else:
            parts.append(rng.normal(0, 0.05, size=shp).astype(np.float32).ravel())
    return np.concatenate(parts)


def make_frames(n, seed=0, density=0.13):
    """Breakout-like sparse uint8 frame stacks [n,4,84,84] (~87% zeros)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(1, 256, size=(n, 4, 84, 84), dtype=np.int64)
    mask = rng.random((n, 4, 84, 84)) < density
    return (x * mask).astype(np.uint8)
