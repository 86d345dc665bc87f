"""parallel_vit.ViT (SURVEY.md section 8 "next" row f4: architectural siblings that reuse the hot-path kernels): every layer is a sum
of parallel attention blocks and a sum of parallel feed-forward blocks (parallel_vit.py:36-42,99-117).  CPU tier: parameter order of
the C library against the oracle's spec, the oracle against a hand-rolled sum of ordinary blocks and finite differences.  GPU tier:
logits and every gradient against the oracle in both compute modes."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ref_torch, spec
from vit_tensorflow import _native as N

KW = dict(image_size=32, patch_size=8, num_classes=7, dim=32, depth=2, heads=2, mlp_dim=64, dim_head=16)
KW_BF16 = dict(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256, dim_head=64)


def _native_cfg(kw, branches):
    c = N.Config()
    c.variant = N.VARIANT_VIT
    c.image_h = c.image_w = kw["image_size"]
    c.patch_h = c.patch_w = kw["patch_size"]
    c.channels, c.num_classes, c.dim, c.depth, c.heads, c.dim_head, c.mlp_dim = 3, kw["num_classes"], kw["dim"], kw["depth"], kw["heads"], kw["dim_head"], kw["mlp_dim"]
    c.num_parallel_branches = branches
    return c


@pytest.mark.parametrize("branches", [2, 3])
def test_param_table_matches_oracle_spec(branches):
    cfg = spec.make_config("vit", **KW, num_parallel_branches=branches)
    tab, n = N.param_table(_native_cfg(KW, branches))
    ps = spec.param_spec(cfg)
    assert [(a, tuple(b)) for a, b, _ in tab] == [(a, tuple(b)) for a, b, _ in ps]
    assert n == sum(int(np.prod(s)) for _, s, _ in ps)
    names = [a for a, _, _ in tab]
    l0 = [x for x in names if x.startswith("transformer.0.")]
    # attribute order of parallel_vit.Transformer: all attention branches of a layer, then all feed-forward branches (parallel_vit.py:106-111)
    assert l0[0] == "transformer.0.attn.0.norm.gamma" and l0.index("transformer.0.mlp.0.norm.gamma") > l0.index(f"transformer.0.attn.{branches - 1}.to_out.bias")
    # one branch is the ordinary ViT (other names): the C library says so by producing the plain table
    plain, _ = N.param_table(_native_cfg(KW, 1))
    assert any(a == "transformer.0.attn.norm.gamma" for a, _, _ in plain)
    bad = _native_cfg(KW, 2)
    bad.variant = N.VARIANT_CAIT
    with pytest.raises(N.VitxError, match="num_parallel_branches"):
        N.param_table(bad)


def test_oracle_is_a_sum_of_ordinary_branches():
    cfg = spec.make_config("vit", **KW, num_parallel_branches=2)
    P = spec.init_params(cfg, 3, randomize_all=True)
    Pt = ref_torch.to_torch(P)
    x = torch.tensor(np.random.default_rng(1).standard_normal((2, 5, 32)))
    got = ref_torch._transformer(x, Pt, cfg, "transformer", 1, ref_torch._ident)
    # hand-rolled: run each branch as an ordinary single-branch block on the same input and add the branch outputs (minus the residual)
    one = spec.make_config("vit", **KW)

    def branch_params(kind, i):
        out = {}
        for k, v in Pt.items():
            if k.startswith(f"transformer.0.{kind}.{i}."):
                out[k.replace(f"transformer.0.{kind}.{i}.", f"transformer.0.{kind}.")] = v
        return out

    acc = x
    for i in range(2):
        Q = branch_params("attn", i)
        y = ref_torch.layer_norm(x, Q["transformer.0.attn.norm.gamma"], Q["transformer.0.attn.norm.beta"])
        acc = acc + ref_torch._attention(y, Q, "transformer.0.attn", one, ref_torch._ident)
    x1 = acc
    acc = x1
    for i in range(2):
        Q = branch_params("mlp", i)
        y = ref_torch.layer_norm(x1, Q["transformer.0.mlp.norm.gamma"], Q["transformer.0.mlp.norm.beta"])
        acc = acc + ref_torch._dense(ref_torch.gelu(ref_torch._dense(y, Q, "transformer.0.mlp.fc1", ref_torch._ident)), Q, "transformer.0.mlp.fc2", ref_torch._ident)
    assert np.abs(got.numpy() - acc.numpy()).max() < 1e-12


def test_oracle_gradients_against_finite_differences():
    cfg = spec.make_config("vit", **KW, num_parallel_branches=2)
    P = spec.init_params(cfg, 4, randomize_all=True)
    rng = np.random.default_rng(2)
    img = rng.standard_normal((2, 32, 32, 3))
    dl = rng.standard_normal((2, 7))
    _, grads, _ = ref_torch.forward_backward(cfg, P, img, dl)
    f = lambda Pp: float((ref_torch.forward(cfg, ref_torch.to_torch(Pp), torch.tensor(img)).numpy() * dl).sum())
    for name in ("transformer.1.attn.1.to_qkv.kernel", "transformer.0.mlp.1.norm.gamma", "transformer.0.attn.0.to_out.bias", "pos_embedding"):
        dirn = rng.standard_normal(P[name].shape)
        eps = 1e-5
        plus, minus = dict(P), dict(P)
        plus[name] = P[name] + eps * dirn
        minus[name] = P[name] - eps * dirn
        fd = (f(plus) - f(minus)) / (2 * eps)
        assert abs(fd - float((grads[name] * dirn).sum())) <= 1e-6 * max(1.0, abs(fd)), name


@pytest.mark.gpu
@pytest.mark.parametrize("compute,branches,pool", [("fp32", 2, "cls"), ("fp32", 3, "mean"), ("bf16", 2, "cls")])
def test_parallel_vit_matches_the_oracle(compute, branches, pool):
    from vit_tensorflow.parallel_vit import ViT
    kw = dict(KW if compute == "fp32" else KW_BF16, pool=pool)
    cfg = spec.make_config("vit", **kw, num_parallel_branches=branches)
    P = spec.init_params(cfg, 5, randomize_all=True)
    b = 3
    m = ViT(**kw, num_parallel_branches=branches, compute=compute, max_batch=b, seed=0)
    assert [n for n, _, _ in m._table] == [n for n, _, _ in spec.param_spec(cfg)]
    m.load_state_dict({k: np.asarray(v, np.float32) for k, v in P.items()})
    rng = np.random.default_rng(6)
    img = rng.standard_normal((b, kw["image_size"], kw["image_size"], 3)).astype(np.float32)
    dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
    logits = m(img, training=False)
    grads, dimg = m.backward(dl, want_dimg=True)
    q = ref_torch.bf16_round if compute == "bf16" else None
    rl, rg, rdimg = ref_torch.forward_backward(cfg, P, img, dl, q=q, want_dimg=True)
    ltol, gtol = (1e-4, 1e-4) if compute == "fp32" else (3e-2, 6e-2)   # bf16: against the oracle with the same rounding points
    assert np.abs(logits - rl).max() <= ltol * max(1.0, np.abs(rl).max())
    for k, r in rg.items():
        assert np.abs(grads[k] - r).max() <= gtol * max(1e-6, np.abs(r).max()) + 1e-7, k
    assert np.abs(dimg - rdimg).max() <= gtol * max(1e-6, np.abs(rdimg).max()) + 1e-7
    # bit-reproducible, and encoder.transformer(tokens) walks the same half-blocks
    l2 = m(img, training=False)
    g2, _ = m.backward(dl)
    assert np.array_equal(logits, l2) and all(np.array_equal(grads[k], g2[k]) for k in grads)
    tok = rng.standard_normal((b, 9, kw["dim"])).astype(np.float32)
    out = m.transformer(tok, training=False)
    ref = ref_torch._transformer(torch.tensor(tok.astype(np.float64)), ref_torch.to_torch(P), cfg, "transformer", cfg["depth"], q or ref_torch._ident).numpy()
    assert np.abs(out - ref).max() <= ltol * max(1.0, np.abs(ref).max())
