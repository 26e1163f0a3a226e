"""IMPALA-ResNet torso (ppo:149-189): C oracle vs torch-CPU float64 autograd."""
import numpy as np
import pytest

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


@pytest.fixture
def hidden_oracle(oracle):
    yield oracle
    oracle.set_resnet_hidden(256)


def test_resnet_hidden_width_layout(hidden_oracle):
    """Network.hiddens = (H,), ppo:94: the oracle, the host layout and the C-ABI count agree for every width the torso takes."""
    import cleanba_amd.lib as L
    import cleanba_amd.model as M
    for hid in (64, 128, 256, 448, 512):
        hidden_oracle.set_resnet_hidden(hid)
        n = hidden_oracle.resnet_param_count(A)
        assert n == 1089232 - 3872 * 256 - 256 + hid * (3872 + 1 + A + 1) + A + 1
        assert n == M.resnet_layout(A, hid)[1] == L.param_count(L.NET_IMPALA_RESNET, A, hid) and M.resnet_hidden_of(n, A) == hid
    keys = [np.array([0, i], np.uint32) for i in range(3)]
    p = M.init_resnet_params(A, *keys, hidden=128)
    q = M.resnet_flax_tree_to_params(M.resnet_params_to_flax_tree(p, A), A)   # checkpoint tree round trip at a non-default width
    assert p.size == M.resnet_layout(A, 128)[1] and (p.view(np.uint32) == q.view(np.uint32)).all()


@pytest.mark.parametrize("hid", [256, 64])
def test_resnet_forward_and_grads_match_torch(hidden_oracle, hid):
    oracle = hidden_oracle
    oracle.set_resnet_hidden(hid)
    rng = np.random.default_rng(1)
    P = make_resnet_params(oracle, 2)
    N = 5
    obs = make_frames(8, 3)
    idx = np.array([6, 1, 4, 0, 7], np.int32)
    logits, value, acts = oracle.resnet_forward(P, A, obs, idx=idx, save_acts=True)
    flat, Pt = tr.unpack_resnet(P, A, requires_grad=True, hid=hid)
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
