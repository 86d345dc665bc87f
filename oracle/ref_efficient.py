"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of two "next"-row siblings of SURVEY.md section 8 (f4).

* efficient.ViT (/root/reference/vit_tensorflow/efficient.py:12-56): patch embedding + cls token + position embedding, a
  transformer object supplied by the caller, pooling + mlp_head.
* the T2T tokenizer (/root/reference/vit_tensorflow/t2t.py:39-47): tf.image.extract_patches(sizes k, strides s, rates 1,
  padding 'SAME') between two rearranges.

Pinned through the reference's own source run under oracle/tf_shim: efficient.py with the reference's vit.Transformer in the middle
(tests/golden/ref_efficient_vit*.npz) is reproduced by the ViT restatement this shell is the depth-0 case of, and t2t.py
(tests/golden/ref_t2t_*.npz; the shim's extract_patches is an independent pad + unfold formulation) by oracle/ref_t2t.py, which
calls extract_patches below.  extract_patches follows TensorFlow's documented SAME rule (out = ceil(in / stride); pad_total = max((out - 1) * stride + k - in, 0);
pad_before = pad_total // 2, the odd pixel goes after; taps outside the image read 0; output depth ordered (row, col, channel));
the loop form below is checked against an independent formulation (explicit zero padding + torch.nn.functional.unfold) in
tests/test_efficient_t2t.py.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import ref_torch


def same_padding(size: int, k: int, stride: int):
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2, total - total // 2


def extract_patches(x: np.ndarray, k: int, stride: int) -> np.ndarray:
    """t2t.py:42 -- plain loops over the window taps (small cases only)."""
    b, H, W, C = x.shape
    oh, pt, _ = same_padding(H, k, stride)
    ow, pl, _ = same_padding(W, k, stride)
    out = np.zeros((b, oh, ow, k, k, C), dtype=x.dtype)
    for oi in range(oh):
        for oj in range(ow):
            for ki in range(k):
                for kj in range(k):
                    y, xx = oi * stride - pt + ki, oj * stride - pl + kj
                    if 0 <= y < H and 0 <= xx < W:
                        out[:, oi, oj, ki, kj, :] = x[:, y, xx, :]
    return out.reshape(b, oh, ow, k * k * C)


def extract_patches_unfold(x: torch.Tensor, k: int, stride: int) -> torch.Tensor:
    """The independent formulation (differentiable): explicit asymmetric zero padding, then im2col."""
    b, H, W, C = x.shape
    oh, pt, pb = same_padding(H, k, stride)
    ow, pl, pr = same_padding(W, k, stride)
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))                 # NCHW, zeros
    cols = torch.nn.functional.unfold(xp, kernel_size=k, stride=stride)                   # [b, C*k*k, L], feature order (c, ki, kj)
    cols = cols.reshape(b, C, k, k, oh, ow).permute(0, 4, 5, 2, 3, 1)                     # -> (ki, kj, c)
    return cols.reshape(b, oh, ow, k * k * C)


def rearrange_unfold(x: np.ndarray, is_first: bool, k: int, stride: int) -> np.ndarray:
    """RearrangeUnfoldTransformer.call without its transformer (t2t.py:39-43)."""
    if not is_first:
        h = int(math.sqrt(x.shape[1]))                                                     # t2t.py:41
        x = x.reshape(x.shape[0], h, x.shape[1] // h, x.shape[2])
    y = extract_patches(x, k, stride)
    return y.reshape(y.shape[0], y.shape[1] * y.shape[2], y.shape[3])


def shell_forward(cfg: dict, P: Dict[str, torch.Tensor], img: torch.Tensor, transformer: Callable[[torch.Tensor], torch.Tensor],
                  q: Optional[Callable] = None) -> torch.Tensor:
    """efficient.ViT.call (efficient.py:38-56); `q` marks the engine's bf16 rounding points as in ref_torch.forward."""
    q = q or (lambda t: t)
    ph, pw = cfg["patch_size"]
    x = ref_torch._dense(q(ref_torch.patch_unfold(img, ph, pw)), P, "patch_embedding", q)   # efficient.py:40
    b, n, d = x.shape
    cls = P["cls_token"].expand(b, 1, d)                                                   # efficient.py:43
    x = torch.cat([cls, x], dim=1) + P["pos_embedding"][:, :n + 1]                         # efficient.py:44-45
    x = transformer(x)                                                                     # efficient.py:46
    x = x.mean(dim=1) if cfg["pool"] == "mean" else x[:, 0]                                # efficient.py:48-51
    x = q(ref_torch.layer_norm(x, P["mlp_head.norm.gamma"], P["mlp_head.norm.beta"]))      # efficient.py:53
    return ref_torch._dense(x, P, "mlp_head", q)


def shell_forward_backward(cfg, params, img, dlogits, transformer, dtype=torch.float64, q=None):
    """Returns (logits, {name: grad} of the shell's parameters, dimg, tokens fed to the transformer)."""
    P = ref_torch.to_torch(params, dtype, requires_grad=True)
    x = torch.tensor(img, dtype=dtype, requires_grad=True)
    seen = {}

    def tf_(t):
        seen["tokens"] = t.detach().numpy().copy()
        return transformer(t)

    logits = shell_forward(cfg, P, x, tf_, q)
    logits.backward(torch.tensor(dlogits, dtype=dtype))
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in P.items()}
    return logits.detach().numpy(), grads, x.grad.numpy(), seen["tokens"]
