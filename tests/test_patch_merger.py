"""vit_with_patch_merger.ViT (SURVEY.md section 8 "next" row f4): no cls token, mean pooling, PatchMerger after the middle layer
(vit_with_patch_merger.py:42-55,112-183).  CPU tier: C parameter table = oracle spec, merger index arithmetic, oracle finite
differences.  GPU tier: logits, every gradient and d(img) against the oracle in both compute modes."""
import numpy as np
import pytest
import torch

from oracle import ref_torch, spec
from vit_tensorflow import _native as N

KW = dict(image_size=32, patch_size=8, num_classes=7, dim=32, depth=4, heads=2, mlp_dim=64, dim_head=16)
KW_BF16 = dict(image_size=64, patch_size=8, num_classes=10, dim=128, depth=4, heads=2, mlp_dim=256, dim_head=64)


def _native_cfg(kw, layer, tokens):
    c = N.Config()
    c.variant = N.VARIANT_PATCH_MERGER
    c.image_h = c.image_w = kw["image_size"]
    c.patch_h = c.patch_w = kw["patch_size"]
    c.channels, c.num_classes, c.dim, c.depth, c.heads, c.dim_head, c.mlp_dim = 3, kw["num_classes"], kw["dim"], kw["depth"], kw["heads"], kw["dim_head"], kw["mlp_dim"]
    c.patch_merge_layer, c.patch_merge_num_tokens = layer or 0, tokens
    return c


def test_param_table_and_merge_index():
    cfg = spec.make_config("patch_merger", **KW, patch_merge_num_tokens=5)
    assert cfg["patch_merge_index"] == 1                                   # depth // 2 - 1
    assert spec.make_config("patch_merger", **KW, patch_merge_layer=4)["patch_merge_index"] == 3
    tab, n = N.param_table(_native_cfg(KW, None, 5))
    ps = spec.param_spec(cfg)
    assert [(a, tuple(b)) for a, b, _ in tab] == [(a, tuple(b)) for a, b, _ in ps]
    names = [a for a, _, _ in tab]
    assert "cls_token" not in names and names.index("transformer.patch_merger.queries") < names.index("transformer.0.attn.norm.gamma")
    assert dict((a, b) for a, b, _ in tab)["transformer.patch_merger.queries"] == (5, 32) and dict((a, b) for a, b, _ in tab)["pos_embedding"] == (1, 17, 32)


def test_oracle_merger_and_finite_differences():
    cfg = spec.make_config("patch_merger", **KW, patch_merge_num_tokens=3)
    P = spec.init_params(cfg, 2, randomize_all=True)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 16, 32))
    Pt = ref_torch.to_torch(P)
    y = ref_torch.patch_merger(torch.tensor(x), Pt, "transformer.patch_merger", 32).numpy()
    # explicit loops: one softmax-weighted average of the normalised tokens per learned query
    g, be, qs = P["transformer.patch_merger.norm.gamma"], P["transformer.patch_merger.norm.beta"], P["transformer.patch_merger.queries"]
    for bi in range(2):
        xn = (x[bi] - x[bi].mean(-1, keepdims=True)) / np.sqrt(x[bi].var(-1, keepdims=True) + 1e-3) * g + be
        for t in range(3):
            s = xn @ qs[t] * 32 ** -0.5
            w = np.exp(s - s.max())
            w /= w.sum()
            assert np.abs(y[bi, t] - w @ xn).max() < 1e-12
    img = rng.standard_normal((2, 32, 32, 3))
    dl = rng.standard_normal((2, 7))
    logits, grads, _ = ref_torch.forward_backward(cfg, P, img, dl)
    assert logits.shape == (2, 7) and not grads["pos_embedding"][0, 16].any() and grads["pos_embedding"][0, :16].any()
    f = lambda Pp: float((ref_torch.forward(cfg, ref_torch.to_torch(Pp), torch.tensor(img)).numpy() * dl).sum())
    for name in ("transformer.patch_merger.queries", "transformer.patch_merger.norm.gamma", "transformer.0.attn.to_qkv.kernel", "transformer.3.mlp.fc2.kernel"):
        dirn = rng.standard_normal(P[name].shape)
        eps = 1e-5
        plus, minus = dict(P), dict(P)
        plus[name] = P[name] + eps * dirn
        minus[name] = P[name] - eps * dirn
        fd = (f(plus) - f(minus)) / (2 * eps)
        assert abs(fd - float((grads[name] * dirn).sum())) <= 1e-6 * max(1.0, abs(fd)), name


@pytest.mark.gpu
@pytest.mark.parametrize("compute,layer,tokens", [("fp32", None, 5), ("fp32", 4, 3), ("fp32", 1, 8), ("bf16", None, 8)])
def test_patch_merger_vit_matches_the_oracle(compute, layer, tokens):
    from vit_tensorflow.vit_with_patch_merger import ViT
    kw = KW if compute == "fp32" else KW_BF16
    cfg = spec.make_config("patch_merger", **kw, patch_merge_layer=layer, patch_merge_num_tokens=tokens)
    P = spec.init_params(cfg, 5, randomize_all=True)
    b = 3
    m = ViT(**kw, patch_merge_layer=layer, patch_merge_num_tokens=tokens, compute=compute, max_batch=b, seed=0)
    assert [n for n, _, _ in m._table] == [n for n, _, _ in spec.param_spec(cfg)] and m.patch_merge_layer_index == cfg["patch_merge_index"]
    m.load_state_dict({k: np.asarray(v, np.float32) for k, v in P.items()})
    rng = np.random.default_rng(6)
    img = rng.standard_normal((b, kw["image_size"], kw["image_size"], 3)).astype(np.float32)
    dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
    q = ref_torch.bf16_round if compute == "bf16" else None
    rl, rg, rdimg = ref_torch.forward_backward(cfg, P, img, dl, q=q, want_dimg=True)
    ltol, gtol = (1e-4, 1e-4) if compute == "fp32" else (3e-2, 6e-2)   # bf16: against the oracle with the same rounding points
    for rep in range(2):                                               # twice: the step-to-step switch of the row extent (n <-> merged tokens)
        logits = m(img, training=False)
        grads, dimg = m.backward(dl, want_dimg=True)
        assert np.abs(logits - rl).max() <= ltol * max(1.0, np.abs(rl).max())
        for k, r in rg.items():
            assert np.abs(grads[k] - r).max() <= gtol * max(1e-6, np.abs(r).max()) + 1e-7, (rep, k)
        assert np.abs(dimg - rdimg).max() <= gtol * max(1e-6, np.abs(rdimg).max()) + 1e-7
    with pytest.raises(N.VitxError, match="ViT / DeepViT only"):
        m.transformer(np.zeros((b, 4, kw["dim"]), np.float32))
