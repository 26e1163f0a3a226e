"""IMPALA-ResNet torso (ppo:149-189): C oracle vs torch-CPU float64 autograd."""
import numpy as np

from helpers import make_frames
import torch_ref as tr

A = 18


def make_resnet_params(oracle, seed):
    rng = np.random.default_rng(seed)
    p = np.zeros(oracle.resnet_param_count(A), np.float32)
    for name, (o, shp) in oracle.resnet_layout(A).items():
        n = int(np.prod(shp))
        if name.endswith(".w"):
            fan = int(np.prod(shp[:-1]))
            p[o:o + n] = rng.normal(0, np.sqrt(1.5 / fan), n)
        else:
            p[o:o + n] = rng.normal(0, 0.05, n)
    return p


def test_resnet_param_count(oracle):
    assert oracle.resnet_param_count(18) == 1094115           # ResNet, A=18
    assert oracle.resnet_param_count(18) - (256 * 18 + 18 + 256 + 1) == 1089232


def test_resnet_forward_and_grads_match_torch(oracle):
    rng = np.random.default_rng(1)
    P = make_resnet_params(oracle, 2)
    N = 5
    obs = make_frames(8, 3)
    idx = np.array([6, 1, 4, 0, 7], np.int32)
    logits, value, acts = oracle.resnet_forward(P, A, obs, idx=idx, save_acts=True)
    flat, Pt = tr.unpack_resnet(P, A, requires_grad=True)
    lt, vt = tr.resnet_forward(Pt, obs[idx])
    scale = float(np.abs(lt.detach().numpy()).max())
    np.testing.assert_allclose(logits, lt.detach().numpy(), rtol=0, atol=2e-6 * scale)
    np.testing.assert_allclose(value, vt.detach().numpy(), rtol=0, atol=2e-6 * scale)
    l14, _ = oracle.resnet_forward(P, A, obs, idx=idx, ksplit=11)
    np.testing.assert_allclose(l14, logits, rtol=0, atol=2e-6 * scale)
    actions = rng.integers(0, A, N).astype(np.int32)
    old_lp = (-np.log(A) + 0.2 * rng.normal(size=N)).astype(np.float32)
    adv = rng.normal(size=N).astype(np.float32)
    tgt = rng.normal(size=N).astype(np.float32)
    stats, dlog, dval = oracle.ppo_loss_head(logits, value, actions, old_lp, adv, tgt)
    grads = oracle.resnet_backward(P, A, obs, idx, acts, dlog, dval)
    loss, _ = tr.ppo_loss(lt, vt, actions, old_lp, adv, tgt)
    loss.backward()
    g = flat.grad.numpy()
    assert abs(stats[0] - loss.item()) < 1e-5
    for name, (o, shp) in oracle.resnet_layout(A).items():
        n = int(np.prod(shp))
        ga, gb = grads[o:o + n], g[o:o + n]
        assert np.abs(ga - gb).max() <= 2e-5 * max(np.abs(gb).max(), 1e-8), (name, np.abs(ga - gb).max(), np.abs(gb).max())
