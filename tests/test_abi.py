"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/cleanba_mi.h declares (no compute calls: there is no GPU here)."""
import os
import re

import numpy as np
import pytest

import cleanba_amd.lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.SO_PATH):
        L.build()
    return L.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "cleanba_mi.h")).read()
    declared = set(re.findall(r"\b(cbm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in cleanba_mi.h but not exported"
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)


def test_config_struct_matches_defaults(lib):
    ppo = L.default_config(L.ALGO_PPO)
    assert (ppo.num_steps, ppo.update_epochs, ppo.num_minibatches, ppo.norm_adv) == (128, 4, 4, 1)
    assert abs(ppo.clip_coef - 0.1) < 1e-7 and abs(ppo.max_grad_norm - 0.5) < 1e-7 and abs(ppo.adam_eps - 1e-5) < 1e-12
    imp = L.default_config(L.ALGO_IMPALA)
    assert (imp.num_steps, imp.update_epochs) == (20, 1) and abs(imp.max_grad_norm - 40.0) < 1e-6
    assert abs(imp.rms_decay - 0.99) < 1e-7 and abs(imp.rms_eps - 0.01) < 1e-8
    assert L.param_count(L.NET_NATURE, 18) == 1693875 and L.param_count(L.NET_NATURE, 4) == 1686693


def test_channels_and_hiddens_travel_in_the_config_and_are_checked_by_the_library(lib):
    """ppo:92-95: the C ABI carries Network(channels, hiddens); widths the HIP torso is not compiled for are refused by cbm_ctx_create (for a C
    host exactly as for the CLI), before any device is touched."""
    import ctypes as C
    assert lib.cbm_config_size() == C.sizeof(L.Config)
    cfg = L.default_config(L.ALGO_PPO)
    assert (cfg.num_channels, list(cfg.channels)[:3], cfg.num_hiddens, cfg.hiddens[0]) == (3, [16, 32, 32], 1, 256)
    cfg.network = L.NET_IMPALA_RESNET
    for mutate, msg in ((lambda c: c.channels.__setitem__(2, 64), "--channels: only"), (lambda c: setattr(c, "num_channels", 2), "--channels: only"),
                        (lambda c: c.hiddens.__setitem__(0, 100), "--hiddens: the HIP"), (lambda c: c.hiddens.__setitem__(0, 576), "--hiddens: the HIP"),
                        (lambda c: setattr(c, "num_hiddens", 2), "--hiddens: the HIP")):
        bad = L.default_config(L.ALGO_PPO)
        bad.network = L.NET_IMPALA_RESNET
        mutate(bad)
        with pytest.raises(L.CbmError, match=msg):
            L.Context(bad)
    from cleanba_amd.args import parse_args
    from cleanba_amd.trainer import make_config
    args = parse_args(["--channels", "16", "32", "64", "--hiddens", "512"], "ppo")
    c2 = make_config(args, "ppo")
    assert list(c2.channels)[:3] == [16, 32, 64] and c2.hiddens[0] == 512
    with pytest.raises(L.CbmError, match="--channels: only"):
        L.Context(c2)


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = L.default_config(L.ALGO_PPO)
    with pytest.raises(L.CbmError):
        L.Context(cfg)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cleanba_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "cbm_oracle" not in txt, f
