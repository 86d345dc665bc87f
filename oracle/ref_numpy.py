"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Pinned against the reference's own source run under oracle/tf_shim (tests/golden/ref_*.npz, tests/test_ref_fixtures.py).

fp64 numpy restatement, op for op, of the reference's forward pass:
  vit_tensorflow/vit.py:14-177, deepvit.py:46-157, cait.py:17-194.
Every rearrange goes through the real `einops` (the library the reference itself calls).
Keras defaults encoded explicitly (Keras source is not under /root/reference):
  LayerNormalization(axis=-1, epsilon=1e-3), biased variance; Dense = x @ kernel[in,out] + bias;
  Softmax(axis=-1); Dropout identity when rate == 0 or training == False (parity setting).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
from einops import rearrange, repeat

try:  # scipy.special.erf is exact to fp64 rounding; fall back to math.erf (vectorised)
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

LN_EPS = 1e-3  # Keras LayerNormalization default epsilon


def layer_norm(x, gamma, beta, eps=LN_EPS):
    """nn.LayerNormalization()  vit.py:18,155"""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def gelu(x):
    """vit.py:34 -- exact erf GELU (the `approximate` branch vit.py:30-32 is dead code)."""
    return 0.5 * x * (1.0 + _erf(x / 1.4142135623730951))


def softmax(x):
    """nn.Softmax() axis=-1  vit.py:58"""
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=-1, keepdims=True)


def patch_unfold(img, ph, pw):
    """Rearrange('b (h p1) (w p2) c -> b (h w) (p1 p2 c)')  vit.py:142"""
    return rearrange(img, 'b (h p1) (w p2) c -> b (h w) (p1 p2 c)', p1=ph, p2=pw)


def mlp(x, P, pre):
    """MLP.net  vit.py:38-44"""
    h = x @ P[f"{pre}.fc1.kernel"] + P[f"{pre}.fc1.bias"]
    h = gelu(h)
    return h @ P[f"{pre}.fc2.kernel"] + P[f"{pre}.fc2.bias"]


def attention_vit(x, P, pre, heads, dim_head, acts=None):
    """Attention.call  vit.py:71-85"""
    qkv = x @ P[f"{pre}.to_qkv.kernel"]                                  # vit.py:72
    q, k, v = np.split(qkv, 3, axis=-1)                                  # vit.py:73
    q, k, v = (rearrange(t, 'b n (h d) -> b h n d', h=heads) for t in (q, k, v))  # vit.py:74
    dots = np.einsum('bhid,bhjd->bhij', q, k) * dim_head ** -0.5         # vit.py:77
    attn = softmax(dots)                                                 # vit.py:78
    out = np.einsum('bhij,bhjd->bhid', attn, v)                          # vit.py:81
    out = rearrange(out, 'b h n d -> b n (h d)')                         # vit.py:82
    if acts is not None:
        acts[f"{pre}.attn_out"] = out
    if f"{pre}.to_out.kernel" in P:                                      # vit.py:53,61-69
        out = out @ P[f"{pre}.to_out.kernel"] + P[f"{pre}.to_out.bias"]
    return out


def attention_deepvit(x, P, pre, heads, dim_head, acts=None):
    """Attention.call  deepvit.py:73-91"""
    qkv = x @ P[f"{pre}.to_qkv.kernel"]
    q, k, v = np.split(qkv, 3, axis=-1)
    q, k, v = (rearrange(t, 'b n (h d) -> b h n d', h=heads) for t in (q, k, v))
    dots = (q @ np.swapaxes(k, -1, -2)) * dim_head ** -0.5               # deepvit.py:79
    attn = softmax(dots)                                                 # deepvit.py:80
    attn = np.einsum('bhij,hg->bgij', attn, P[f"{pre}.reattn_weights"])  # deepvit.py:83
    attn = rearrange(attn, 'b h i j -> b i j h')                         # deepvit.py:60
    attn = layer_norm(attn, P[f"{pre}.reattn_norm.gamma"], P[f"{pre}.reattn_norm.beta"])  # :61
    attn = rearrange(attn, 'b i j h -> b h i j')                         # deepvit.py:62
    out = attn @ v                                                       # deepvit.py:87
    out = rearrange(out, 'b h n d -> b n (h d)')
    if acts is not None:
        acts[f"{pre}.attn_out"] = out
    return out @ P[f"{pre}.to_out.kernel"] + P[f"{pre}.to_out.bias"]     # deepvit.py:65-68 unconditional


def attention_cait(x, P, pre, heads, dim_head, context=None, acts=None):
    """Attention.call  cait.py:107-131.  `x` is already LayerNormed by PreNorm; `context`
    arrives through **kwargs un-normalised (cait.py:57-58)."""
    ctx = x if context is None else np.concatenate([x, context], axis=1)  # cait.py:109-112
    q = x @ P[f"{pre}.to_q.kernel"]                                      # cait.py:114
    kv = ctx @ P[f"{pre}.to_kv.kernel"]                                  # cait.py:115
    k, v = np.split(kv, 2, axis=-1)                                      # cait.py:116
    q, k, v = (rearrange(t, 'b n (h d) -> b h n d', h=heads) for t in (q, k, v))
    dots = np.einsum('bhid,bhjd->bhij', q, k) * dim_head ** -0.5         # cait.py:121
    dots = np.einsum('bhij,hg->bgij', dots, P[f"{pre}.mix_heads_pre_attn"])   # cait.py:123
    attn = softmax(dots)                                                 # cait.py:124
    attn = np.einsum('bhij,hg->bgij', attn, P[f"{pre}.mix_heads_post_attn"])  # cait.py:125
    out = attn @ v                                                       # cait.py:127
    out = rearrange(out, 'b h n d -> b n (h d)')
    if acts is not None:
        acts[f"{pre}.attn_out"] = out
    return out @ P[f"{pre}.to_out.kernel"] + P[f"{pre}.to_out.bias"]     # cait.py:129


def transformer(x, P, cfg, prefix, depth, acts=None, context=None):
    """Transformer.call  vit.py:99-104 / deepvit.py:106-110 / cait.py:146-153 (layer_dropout = 0)."""
    v, h, dh = cfg["variant"], cfg["heads"], cfg["dim_head"]
    for i in range(depth):
        pa, pm = f"{prefix}.{i}.attn", f"{prefix}.{i}.mlp"
        xn = layer_norm(x, P[f"{pa}.norm.gamma"], P[f"{pa}.norm.beta"])  # PreNorm vit.py:22
        if v == "vit":
            a = attention_vit(xn, P, pa, h, dh, acts)
        elif v == "deepvit":
            a = attention_deepvit(xn, P, pa, h, dh, acts)
        else:
            a = attention_cait(xn, P, pa, h, dh, context=context, acts=acts) * P[f"{pa}.scale"]  # cait.py:47-48
        x = a + x                                                        # vit.py:101
        xn = layer_norm(x, P[f"{pm}.norm.gamma"], P[f"{pm}.norm.beta"])
        f = mlp(xn, P, pm)
        if v == "cait":
            f = f * P[f"{pm}.scale"]
        x = f + x                                                        # vit.py:102
        if acts is not None:
            acts[f"{prefix}.{i}.out"] = x
    return x


def forward(cfg: dict, params: Dict[str, np.ndarray], img: np.ndarray, acts: Optional[dict] = None) -> np.ndarray:
    """ViT.call vit.py:159-177 / DeepViT.call deepvit.py:139-157 / CaiT.call cait.py:180-194.
    img: [b, H, W, 3] (NHWC).  Returns logits [b, num_classes] in fp64."""
    P = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    img = np.asarray(img, dtype=np.float64)
    ph, pw = cfg["patch_size"]
    x = patch_unfold(img, ph, pw)                                        # vit.py:142
    if acts is not None:
        acts["patches"] = x
    x = x @ P["patch_embedding.kernel"] + P["patch_embedding.bias"]      # vit.py:143
    b, n, _ = x.shape                                                    # vit.py:161
    if cfg["variant"] == "cait":
        x = x + P["pos_embedding"][:, :n]                                # cait.py:184
        if acts is not None:
            acts["embed"] = x
        x = transformer(x, P, cfg, "patch_transformer", cfg["depth"], acts)          # cait.py:187
        cls_tokens = repeat(P["cls_token"], '() n d -> b n d', b=b)      # cait.py:189
        x = transformer(cls_tokens, P, cfg, "cls_transformer", cfg["cls_depth"], acts, context=x)  # :190
        x = x[:, 0]                                                      # cait.py:192
    else:
        cls_tokens = repeat(P["cls_token"], '() n d -> b n d', b=b)      # vit.py:163
        x = np.concatenate([cls_tokens, x], axis=1)                      # vit.py:164
        x = x + P["pos_embedding"][:, :(n + 1)]                          # vit.py:165
        if acts is not None:
            acts["embed"] = x
        x = transformer(x, P, cfg, "transformer", cfg["depth"], acts)    # vit.py:168
        x = x.mean(axis=1) if cfg["pool"] == "mean" else x[:, 0]         # vit.py:170-173
    if acts is not None:
        acts["pooled"] = x
    x = layer_norm(x, P["mlp_head.norm.gamma"], P["mlp_head.norm.beta"])  # vit.py:155
    return x @ P["mlp_head.kernel"] + P["mlp_head.bias"]                 # vit.py:156
