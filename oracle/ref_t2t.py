"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

torch (float64, autograd) restatement of T2TViT (vit_tensorflow/t2t.py:49-122): the token-to-token patch embedding -- per layer
[tokens -> square grid ->] tf.image.extract_patches(k, stride, 'SAME') -> tokens [-> one-block Transformer(dim=layer_dim, heads=1,
dim_head=layer_dim, mlp_dim=layer_dim), all but the last layer] (t2t.py:39-47,59-72), then Dense(dim) (t2t.py:74) -- followed by the
ordinary cls / position / transformer / pool / mlp_head path of vit.py (t2t.py:99-121).  Pinned by tests/golden/ref_t2t_*.npz, which
oracle/gen_ref_fixtures.py produces by running the reference's own t2t.py under oracle/tf_shim.

Parameter names: `patch_embedding.{i}.transformer_layer.0.<attn|mlp>...` for the tokenizer's transformers (vit.py:53: heads == 1 and
dim_head == dim, so their attention has no to_out), `patch_embedding.{L}.kernel|bias` for the Dense (L = number of t2t layers: its index
in the reference's Sequential), then `pos_embedding, cls_token, transformer.*, mlp_head.*` as for vit.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import ref_efficient, ref_torch, spec


def conv_output_size(image_size, kernel_size, stride, padding):
    """t2t.py:15-16"""
    return int(((image_size - kernel_size + (2 * padding)) / stride) + 1)


def make_config(image_size, num_classes, dim, depth, heads, mlp_dim, pool="cls", channels=3, dim_head=64,
                t2t_layers=((7, 4), (3, 2), (3, 2)), **_ignored) -> dict:
    assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'      # t2t.py:54
    layer_dims, layer_dim, out_size = [], channels, image_size
    for k, s in t2t_layers:                                                                               # t2t.py:60-66
        layer_dim *= k ** 2
        out_size = conv_output_size(out_size, k, s, s // 2)
        layer_dims.append(layer_dim)
    return dict(image_size=image_size, num_classes=num_classes, dim=dim, depth=depth, heads=heads, mlp_dim=mlp_dim, pool=pool, channels=channels,
                dim_head=dim_head, t2t_layers=tuple(tuple(x) for x in t2t_layers), layer_dims=layer_dims, num_pos=out_size ** 2 + 1)   # t2t.py:77


def _inner_cfg(ld: int) -> dict:
    """Transformer(dim=layer_dim, heads=1, depth=1, dim_head=layer_dim, mlp_dim=layer_dim)  t2t.py:37 (args at t2t.py:69-70)"""
    return dict(variant="vit", dim=ld, depth=1, heads=1, dim_head=ld, mlp_dim=ld, num_parallel_branches=1)


def param_spec(cfg: dict) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Keras' `Model.weights` order for t2t.py:49-122: a layer's OWN variables first (pos_embedding, cls_token: the tf.Variables the model
    assigns to itself, t2t.py:77-78), then its sublayers in attribute order (patch_embedding Sequential, transformer, mlp_head) -- the same rule
    the engine's ViT table follows (pos_embedding, cls_token, patch_embedding.*).  `init_params` draws the random values in `_draw_order`, the
    order of this file's first version, so that the committed reference fixtures (tests/golden/ref_t2t_*.npz) keep their weights."""
    d = _draw_order(cfg)
    own = [e for e in d if e[0] in ("pos_embedding", "cls_token")]
    return own + [e for e in d if e[0] not in ("pos_embedding", "cls_token")]


def _draw_order(cfg: dict) -> List[Tuple[str, Tuple[int, ...], str]]:
    out = []
    add = lambda n, s, k: out.append((n, tuple(s), k))
    L = len(cfg["t2t_layers"])
    for i, ld in enumerate(cfg["layer_dims"][:-1]):
        pre = f"patch_embedding.{i}.transformer_layer.0"
        add(f"{pre}.attn.norm.gamma", (ld,), "ones"); add(f"{pre}.attn.norm.beta", (ld,), "zeros")
        add(f"{pre}.attn.to_qkv.kernel", (ld, 3 * ld), "glorot")
        add(f"{pre}.mlp.norm.gamma", (ld,), "ones"); add(f"{pre}.mlp.norm.beta", (ld,), "zeros")
        add(f"{pre}.mlp.fc1.kernel", (ld, ld), "glorot"); add(f"{pre}.mlp.fc1.bias", (ld,), "zeros")
        add(f"{pre}.mlp.fc2.kernel", (ld, ld), "glorot"); add(f"{pre}.mlp.fc2.bias", (ld,), "zeros")
    add(f"patch_embedding.{L}.kernel", (cfg["layer_dims"][-1], cfg["dim"]), "glorot")
    add(f"patch_embedding.{L}.bias", (cfg["dim"],), "zeros")
    d, h, dh, m, nc = cfg["dim"], cfg["heads"], cfg["dim_head"], cfg["mlp_dim"], cfg["num_classes"]
    add("pos_embedding", (1, cfg["num_pos"], d), "normal")
    add("cls_token", (1, 1, d), "normal")
    for l in range(cfg["depth"]):
        pre = f"transformer.{l}"
        add(f"{pre}.attn.norm.gamma", (d,), "ones"); add(f"{pre}.attn.norm.beta", (d,), "zeros")
        add(f"{pre}.attn.to_qkv.kernel", (d, 3 * h * dh), "glorot")
        if not (h == 1 and dh == d):
            add(f"{pre}.attn.to_out.kernel", (h * dh, d), "glorot"); add(f"{pre}.attn.to_out.bias", (d,), "zeros")
        add(f"{pre}.mlp.norm.gamma", (d,), "ones"); add(f"{pre}.mlp.norm.beta", (d,), "zeros")
        add(f"{pre}.mlp.fc1.kernel", (d, m), "glorot"); add(f"{pre}.mlp.fc1.bias", (m,), "zeros")
        add(f"{pre}.mlp.fc2.kernel", (m, d), "glorot"); add(f"{pre}.mlp.fc2.bias", (d,), "zeros")
    add("mlp_head.norm.gamma", (d,), "ones"); add("mlp_head.norm.beta", (d,), "zeros")
    add("mlp_head.kernel", (d, nc), "glorot"); add("mlp_head.bias", (nc,), "zeros")
    return out


def init_params(cfg: dict, seed: int = 1, randomize_all: bool = True) -> Dict[str, np.ndarray]:
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape, kind in _draw_order(cfg):
        if kind == "normal":
            a = rng.standard_normal(shape)
        elif kind == "glorot":
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-lim, lim, shape)
        elif kind == "zeros":
            a = 0.1 * rng.standard_normal(shape) if randomize_all else np.zeros(shape)
        else:
            a = 1.0 + 0.1 * rng.standard_normal(shape) if randomize_all else np.ones(shape)
        out[name] = np.ascontiguousarray(a, dtype=np.float64)
    return out


def tokenizer(cfg: dict, P: Dict[str, torch.Tensor], img: torch.Tensor) -> torch.Tensor:
    """The Sequential of RearrangeUnfoldTransformer layers (t2t.py:59-72): returns tokens [b, n, last layer_dim]."""
    x = img
    L = len(cfg["t2t_layers"])
    for i, (k, s) in enumerate(cfg["t2t_layers"]):
        if i > 0:                                                                   # t2t.py:40-41
            h = int(math.sqrt(x.shape[1]))
            x = x.reshape(x.shape[0], h, x.shape[1] // h, x.shape[2])
        x = ref_efficient.extract_patches_unfold(x, k, s)                           # t2t.py:42
        x = x.reshape(x.shape[0], x.shape[1] * x.shape[2], x.shape[3])              # t2t.py:43
        if i < L - 1:                                                               # t2t.py:44-45
            sub = {kk[len(f"patch_embedding.{i}.transformer_layer."):]: v for kk, v in P.items() if kk.startswith(f"patch_embedding.{i}.transformer_layer.")}
            x = ref_torch._transformer(x, {f"t.{kk}": v for kk, v in sub.items()}, _inner_cfg(cfg["layer_dims"][i]), "t", 1, ref_torch._ident)
    return x


def forward(cfg: dict, P: Dict[str, torch.Tensor], img: torch.Tensor) -> torch.Tensor:
    """T2TViT.call (t2t.py:99-121)"""
    L = len(cfg["t2t_layers"])
    x = tokenizer(cfg, P, img) @ P[f"patch_embedding.{L}.kernel"] + P[f"patch_embedding.{L}.bias"]      # t2t.py:74,100
    b, n, d = x.shape
    x = torch.cat([P["cls_token"].expand(b, 1, d), x], dim=1) + P["pos_embedding"][:, :n + 1]            # t2t.py:103-105
    vcfg = dict(variant="vit", dim=cfg["dim"], heads=cfg["heads"], dim_head=cfg["dim_head"], mlp_dim=cfg["mlp_dim"], num_parallel_branches=1)
    x = ref_torch._transformer(x, P, vcfg, "transformer", cfg["depth"], ref_torch._ident)               # t2t.py:108
    x = x.mean(dim=1) if cfg["pool"] == "mean" else x[:, 0]                                              # t2t.py:110-113
    x = ref_torch.layer_norm(x, P["mlp_head.norm.gamma"], P["mlp_head.norm.beta"])
    return x @ P["mlp_head.kernel"] + P["mlp_head.bias"]                                                # t2t.py:115


def student_forward(cfg: dict, P: Dict[str, torch.Tensor], img: torch.Tensor, distill_token: torch.Tensor):
    """DistillMixin.call for DistillableT2TViT (distill.py:16-44,60-72): the token joins after the position embedding."""
    L = len(cfg["t2t_layers"])
    x = tokenizer(cfg, P, img) @ P[f"patch_embedding.{L}.kernel"] + P[f"patch_embedding.{L}.bias"]      # distill.py:18
    b, n, d = x.shape
    x = torch.cat([P["cls_token"].expand(b, 1, d), x], dim=1) + P["pos_embedding"][:, :n + 1]            # distill.py:21-23
    x = torch.cat([x, distill_token.reshape(1, 1, d).expand(b, 1, d)], dim=1)                            # distill.py:25-27
    vcfg = dict(variant="vit", dim=cfg["dim"], heads=cfg["heads"], dim_head=cfg["dim_head"], mlp_dim=cfg["mlp_dim"], num_parallel_branches=1)
    x = ref_torch._transformer(x, P, vcfg, "transformer", cfg["depth"], ref_torch._ident)               # distill.py:29
    x, dtok = x[:, :-1], x[:, -1]                                                                        # distill.py:32
    x = x.mean(dim=1) if cfg["pool"] == "mean" else x[:, 0]
    x = ref_torch.layer_norm(x, P["mlp_head.norm.gamma"], P["mlp_head.norm.beta"])
    return x @ P["mlp_head.kernel"] + P["mlp_head.bias"], dtok


def forward_backward(cfg, params, img, dlogits, want_dimg=True):
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    x = torch.tensor(np.asarray(img, np.float64), requires_grad=want_dimg)
    logits = forward(cfg, P, x)
    logits.backward(torch.tensor(np.asarray(dlogits, np.float64)))
    return logits.detach().numpy(), {k: v.grad.numpy() for k, v in P.items()}, (x.grad.numpy() if want_dimg else None)
