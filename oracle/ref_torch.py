"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Pinned against the reference's own source run under oracle/tf_shim (tests/golden/ref_*.npz, tests/test_ref_fixtures.py).

Independent torch-CPU twin of oracle/ref_numpy.py (plain reshape/permute instead of einops)
with autograd, used as the gradient oracle (the reference has no backward code: gradients are
TF autodiff of vit.py:159-177) and, in fp32 with all host threads, as the
"reference-restatement CPU baseline" that bench.py times.

`q` is an optional rounding hook applied wherever the bf16 throughput mode of the engine stores
a bf16 tensor (GEMM operands, saved activations); q=None is the exact oracle.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

LN_EPS = 1e-3


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bf16 and back, straight-through for autograd."""
    r = t.detach().to(torch.bfloat16).to(t.dtype)
    return t + (r - t.detach())


def _ident(t):
    return t


def layer_norm(x, g, b):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + LN_EPS) * g + b


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / 1.4142135623730951))


def patch_unfold(img, ph, pw):
    b, H, W, c = img.shape
    x = img.reshape(b, H // ph, ph, W // pw, pw, c)
    x = x.permute(0, 1, 3, 2, 4, 5)
    return x.reshape(b, (H // ph) * (W // pw), ph * pw * c)


def _heads(t, h):
    b, n, _ = t.shape
    return t.reshape(b, n, h, -1).permute(0, 2, 1, 3)


def _merge(t):
    b, h, n, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, n, h * d)


def _dense(x, P, name, q, bias=True):
    y = q(x) @ q(P[f"{name}.kernel"])
    if bias:
        y = y + P[f"{name}.bias"]
    return y


def _attention(x, P, pre, cfg, q, context=None):
    v, h, dh = cfg["variant"], cfg["heads"], cfg["dim_head"]
    scale = dh ** -0.5
    if v == "cait":
        ctx = x if context is None else torch.cat([x, q(context)], dim=1)
        qq = q(_dense(x, P, f"{pre}.to_q", q, bias=False))
        kv = q(_dense(ctx, P, f"{pre}.to_kv", q, bias=False))
        kk, vv = kv.chunk(2, dim=-1)
    else:
        qkv = q(_dense(x, P, f"{pre}.to_qkv", q, bias=False))
        qq, kk, vv = qkv.chunk(3, dim=-1)
    qq, kk, vv = _heads(qq, h), _heads(kk, h), _heads(vv, h)
    dots = (qq @ kk.transpose(-1, -2)) * scale
    if v == "cait":
        dots = torch.einsum('bhij,hg->bgij', dots, P[f"{pre}.mix_heads_pre_attn"])
    attn = torch.softmax(dots, dim=-1)
    if v == "cait":
        attn = torch.einsum('bhij,hg->bgij', attn, P[f"{pre}.mix_heads_post_attn"])
    elif v == "deepvit":
        attn = torch.einsum('bhij,hg->bgij', attn, P[f"{pre}.reattn_weights"])
        attn = layer_norm(attn.permute(0, 2, 3, 1), P[f"{pre}.reattn_norm.gamma"],
                          P[f"{pre}.reattn_norm.beta"]).permute(0, 3, 1, 2)
    out = q(_merge(q(attn) @ vv))
    if f"{pre}.to_out.kernel" in P:
        out = _dense(out, P, f"{pre}.to_out", q)
    return out


def _parallel_transformer(x, P, cfg, prefix, depth, q):
    """parallel_vit.py:99-117: x = sum_i attn_i(norm_i(x)) + x; x = sum_i ff_i(norm_i(x)) + x (every branch has its own PreNorm)."""
    nb = cfg["num_parallel_branches"]
    for l in range(depth):
        acc = x
        for i in range(nb):                                                         # Parallel.call: sum of the branches  parallel_vit.py:41-42
            pa = f"{prefix}.{l}.attn.{i}"
            y = q(layer_norm(x, P[f"{pa}.norm.gamma"], P[f"{pa}.norm.beta"]))       # PreNorm  parallel_vit.py:44-52
            qkv = q(q(y) @ q(P[f"{pa}.to_qkv.kernel"]))
            qq, kk, vv = (_heads(t, cfg["heads"]) for t in qkv.chunk(3, dim=-1))
            attn = torch.softmax((qq @ kk.transpose(-1, -2)) * cfg["dim_head"] ** -0.5, dim=-1)
            out = q(_merge(q(attn) @ vv))
            if f"{pa}.to_out.kernel" in P:
                out = q(out) @ q(P[f"{pa}.to_out.kernel"]) + P[f"{pa}.to_out.bias"]
            acc = acc + out
        x = acc                                                                     # attns(x) + x  parallel_vit.py:114
        acc = x
        for i in range(nb):
            pm = f"{prefix}.{l}.mlp.{i}"
            y = q(layer_norm(x, P[f"{pm}.norm.gamma"], P[f"{pm}.norm.beta"]))
            hpre = q(q(y) @ q(P[f"{pm}.fc1.kernel"]) + P[f"{pm}.fc1.bias"])
            acc = acc + (q(gelu(hpre)) @ q(P[f"{pm}.fc2.kernel"]) + P[f"{pm}.fc2.bias"])
        x = acc                                                                     # ffs(x) + x  parallel_vit.py:115
    return x


def _transformer(x, P, cfg, prefix, depth, q, context=None):
    if cfg.get("num_parallel_branches", 1) > 1:
        return _parallel_transformer(x, P, cfg, prefix, depth, q)
    cait = cfg["variant"] == "cait"
    merge_at = cfg.get("patch_merge_index", -1) if cfg["variant"] == "patch_merger" else -1
    for i in range(depth):
        pa, pm = f"{prefix}.{i}.attn", f"{prefix}.{i}.mlp"
        a = _attention(q(layer_norm(x, P[f"{pa}.norm.gamma"], P[f"{pa}.norm.beta"])), P, pa, cfg, q, context)
        if cait:
            a = a * P[f"{pa}.scale"]
        x = a + x
        hpre = q(_dense(q(layer_norm(x, P[f"{pm}.norm.gamma"], P[f"{pm}.norm.beta"])), P, f"{pm}.fc1", q))
        f = _dense(q(gelu(hpre)), P, f"{pm}.fc2", q)
        if cait:
            f = f * P[f"{pm}.scale"]
        x = f + x
        if i == merge_at:                                                      # vit_with_patch_merger.py:131-132
            x = patch_merger(x, P, f"{prefix}.patch_merger", cfg["dim"])
    return x


def patch_merger(x, P, pre, dim):
    """PatchMerger.call (vit_with_patch_merger.py:49-55): exact fp32 in both engine modes, so no rounding hook here."""
    xn = layer_norm(x, P[f"{pre}.norm.gamma"], P[f"{pre}.norm.beta"])                       # :50
    sim = P[f"{pre}.queries"] @ (xn.transpose(-1, -2) * dim ** -0.5)                       # :51
    return torch.softmax(sim, dim=-1) @ xn                                                 # :52-53


def forward(cfg: dict, P: Dict[str, torch.Tensor], img: torch.Tensor,
            q: Optional[Callable] = None) -> torch.Tensor:
    q = q or _ident
    ph, pw = cfg["patch_size"]
    x = _dense(q(patch_unfold(img, ph, pw)), P, "patch_embedding", q)
    b, n, d = x.shape
    if cfg["variant"] == "patch_merger":                                       # vit_with_patch_merger.py:173-183
        x = x + P["pos_embedding"][:, :n]
        x = _transformer(x, P, cfg, "transformer", cfg["depth"], q)
        x = q(layer_norm(x.mean(dim=1), P["mlp_head.norm.gamma"], P["mlp_head.norm.beta"]))   # Reduce mean, LayerNorm  :168-171
        return _dense(x, P, "mlp_head", q)
    cls = P["cls_token"].expand(b, 1, d)
    if cfg["variant"] == "cait":
        x = x + P["pos_embedding"][:, :n]
        x = _transformer(x, P, cfg, "patch_transformer", cfg["depth"], q)
        x = _transformer(cls, P, cfg, "cls_transformer", cfg["cls_depth"], q, context=x)
        x = x[:, 0]
    else:
        x = torch.cat([cls, x], dim=1) + P["pos_embedding"][:, :n + 1]
        x = _transformer(x, P, cfg, "transformer", cfg["depth"], q)
        x = x.mean(dim=1) if cfg["pool"] == "mean" else x[:, 0]
    x = q(layer_norm(x, P["mlp_head.norm.gamma"], P["mlp_head.norm.beta"]))
    return _dense(x, P, "mlp_head", q)


def to_torch(params, dtype=torch.float64, requires_grad=False):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in params.items()}


def forward_backward(cfg, params, img, dlogits, dtype=torch.float64, q=None, want_dimg=False):
    """Returns (logits, {name: grad}, dimg|None): VJP of forward() for the cotangent `dlogits`."""
    P = to_torch(params, dtype, requires_grad=True)
    x = torch.tensor(img, dtype=dtype, requires_grad=want_dimg)
    logits = forward(cfg, P, x, q)
    logits.backward(torch.tensor(dlogits, dtype=dtype))
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in P.items()}
    return logits.detach().numpy(), grads, (x.grad.numpy() if want_dimg else None)


def ce_dlogits(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """d/dlogits of mean softmax cross-entropy (the only loss in the reference: distill.py:119)."""
    p = torch.softmax(logits, dim=-1)
    p[torch.arange(logits.shape[0]), labels] -= 1.0
    return p / logits.shape[0]


def transformer_forward_backward(cfg, params, tokens, dout, dtype=torch.float64, q=None):
    """encoder.transformer(tokens) (mae.py:69 / vit.py:92-104) and its VJP for the cotangent `dout`:
    returns (out, {name: grad for the transformer's parameters}, dtokens)."""
    q = q or _ident
    P = to_torch(params, dtype, requires_grad=True)
    x = torch.tensor(tokens, dtype=dtype, requires_grad=True)
    out = _transformer(x, P, cfg, "transformer", cfg["depth"], q)
    out.backward(torch.tensor(dout, dtype=dtype))
    grads = {k: (v.grad.detach().numpy() if v.grad is not None else None) for k, v in P.items()}
    return out.detach().numpy(), grads, x.grad.detach().numpy()
