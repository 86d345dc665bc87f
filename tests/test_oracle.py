"""CPU tier: the oracle against itself (numpy fp64 vs torch fp64 twins), against the committed golden
fixtures, against the real `einops` for the patch-index arithmetic, and finite-difference checks of the
gradient oracle.  (What pins the oracle to the reference itself is tests/test_ref_fixtures.py.)"""
import os

import numpy as np
import pytest
import torch
from einops import rearrange
from hypothesis import given, settings, strategies as st

from oracle import ref_numpy, ref_torch, spec
from util import CONFIGS, oracle_cfg

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
SMALL = ["vit_small", "vit_rect_mean", "vit_noproj", "deepvit_small", "cait_small"]


@pytest.mark.parametrize("name", SMALL)
def test_twins_agree(name):
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, 3, randomize_all=True)
    img = np.random.default_rng(0).standard_normal((2, *cfg["image_size"], 3))
    a = ref_numpy.forward(cfg, P, img)
    b = ref_torch.forward(cfg, ref_torch.to_torch(P), torch.tensor(img)).numpy()
    assert a.shape == (2, cfg["num_classes"])          # the reference's only documented check (vit.py:194)
    assert np.abs(a - b).max() < 1e-12


@pytest.mark.parametrize("name", SMALL)
def test_golden_fixture(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, seed=1, randomize_all=True)
    assert abs(sum(float(np.abs(v).sum()) for v in P.values()) - float(z["param_checksum"])) < 1e-6
    logits = ref_numpy.forward(cfg, P, z["img"])
    assert np.abs(logits - z["logits"]).max() < 1e-10
    _, grads, dimg = ref_torch.forward_backward(cfg, P, z["img"], z["dlogits"], want_dimg=True)
    for k, g in grads.items():
        ref = z["grad/" + k]
        assert np.abs(g - ref).max() <= 1e-6 * (np.abs(ref).max() + 1e-12) + 1e-9, k
    assert np.abs(dimg - z["dimg"]).max() < 1e-9


@settings(max_examples=40, deadline=None)
@given(b=st.integers(0, 3), hp=st.integers(1, 4), wp=st.integers(1, 4), p1=st.integers(1, 5), p2=st.integers(1, 5), c=st.integers(1, 4))
def test_unfold_formula_matches_einops(b, hp, wp, p1, p2, c):
    """P[b,t,f] = img[b, (t div Wp)*p1 + r, (t mod Wp)*p2 + s, c], f = (r*p2+s)*C + c."""
    H, W = hp * p1, wp * p2
    img = np.arange(b * H * W * c, dtype=np.int64).reshape(b, H, W, c)
    ref = rearrange(img, 'b (h p1) (w p2) c -> b (h w) (p1 p2 c)', p1=p1, p2=p2)   # vit.py:142
    out = np.empty_like(ref)
    for t in range(hp * wp):
        for f in range(p1 * p2 * c):
            r, s, cc = f // (p2 * c), (f // c) % p2, f % c
            out[:, t, f] = img[:, (t // wp) * p1 + r, (t % wp) * p2 + s, cc]
    assert np.array_equal(out, ref)
    assert np.array_equal(ref_numpy.patch_unfold(img, p1, p2), ref)
    assert np.array_equal(ref_torch.patch_unfold(torch.tensor(img), p1, p2).numpy(), ref)


@pytest.mark.parametrize("name", ["vit_small", "deepvit_small", "cait_small"])
def test_gradient_oracle_finite_difference(name):
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, 5, randomize_all=True)
    rng = np.random.default_rng(1)
    img = rng.standard_normal((1, *cfg["image_size"], 3))
    dl = rng.standard_normal((1, cfg["num_classes"]))
    _, grads, _ = ref_torch.forward_backward(cfg, P, img, dl)
    names = [n for n, _, _ in spec.param_spec(cfg)]
    for pname in names[:: max(1, len(names) // 8)]:
        idx = tuple(rng.integers(0, s) for s in P[pname].shape)
        eps = 1e-5
        Pp = {k: v.copy() for k, v in P.items()}
        Pm = {k: v.copy() for k, v in P.items()}
        Pp[pname][idx] += eps
        Pm[pname][idx] -= eps
        fd = ((ref_numpy.forward(cfg, Pp, img) - ref_numpy.forward(cfg, Pm, img)) * dl).sum() / (2 * eps)
        assert abs(fd - grads[pname][idx]) <= 1e-5 * max(1.0, abs(fd)), (pname, fd, grads[pname][idx])


def test_keras_semantics_encoded():
    """Facts read off the reference that differ from PyTorch habits."""
    x = np.random.default_rng(0).standard_normal((3, 8))
    y = ref_numpy.layer_norm(x, np.ones(8), np.zeros(8))
    var = x.var(axis=-1, keepdims=True)                       # biased variance, eps = 1e-3
    assert np.allclose(y, (x - x.mean(-1, keepdims=True)) / np.sqrt(var + 1e-3))
    assert abs(ref_numpy.gelu(np.array([1.0]))[0] - 0.8413447460685429) < 1e-12    # exact-erf GELU (vit.py:34)
    cfg = spec.make_config("vit", image_size=32, patch_size=8, num_classes=3, dim=64, depth=1, heads=1, mlp_dim=64, dim_head=64)
    assert not any("to_out" in n for n, _, _ in spec.param_spec(cfg))               # vit.py:53
    cfg = spec.make_config("deepvit", image_size=32, patch_size=8, num_classes=3, dim=64, depth=1, heads=1, mlp_dim=64, dim_head=64)
    assert any("to_out" in n for n, _, _ in spec.param_spec(cfg))                   # deepvit.py:65-68 unconditional
    assert spec.layer_scale_init(18) == 0.1 and spec.layer_scale_init(19) == 1e-5 and spec.layer_scale_init(25) == 1e-6


def test_flop_model_matches_survey():
    cfg = oracle_cfg("cfg2_vit_b16")
    assert abs(spec.flops_per_image(cfg) / 1e9 - 105.383) < 1e-3
    assert sum(int(np.prod(s)) for _, s, _ in spec.param_spec(cfg)) == 86540008
    assert abs(spec.flops_per_image(oracle_cfg("cfg1_readme")) / 1e9 - 21.155) < 2e-3
