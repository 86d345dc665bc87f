"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Pinned: tests/test_ref_fixtures.py checks this module against tests/golden/ref_mae_*.npz and
ref_simmim_vit.npz, produced by the reference's own mae.py / simmim.py under oracle/tf_shim (oracle/gen_ref_fixtures.py:make_mim).

Torch-CPU restatement (fp64 by default, autograd) of the two masked-image-modelling wrappers that call
`encoder.transformer(tokens)` on something other than the full token sequence:

    MAE.call     vit_tensorflow/mae.py:47-92
    SimMIM.call  vit_tensorflow/simmim.py:86-130

The random draws of the reference (tf.random.uniform + argsort, mae.py:58; top_k, simmim.py:108) are inputs here, so the
engine and the oracle can be run on the same indices.  Two places where the reference's code does not do what it evidently
means are restated literally and flagged:

  * mae.py:90  `tf.reduce_mean(tf.square(pred_pixel_values, masked_patches))`: tf.square's second positional parameter is
    `name`, so as written the loss is mean(pred**2).  `literal_loss=True` follows the code, `False` the intended MSE.
  * mae.py:62 / simmim.py:119 index through `.numpy()`, which detaches the result from the GradientTape: in the reference
    nothing upstream of that line receives a gradient.  The forward values are unaffected.  `detach_like_reference=True`
    reproduces the cut (torch .detach()), `False` differentiates through the gather (what the engine implements).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import ref_torch as R


def num_masked(masking_ratio: float, num_patches: int) -> int:
    return int(masking_ratio * num_patches)   # mae.py:57, simmim.py:106


def mae_param_spec(enc_cfg: dict, num_patches: int, decoder_dim: int):
    """(name, shape) of the wrapper's own parameters, in the attribute order of MAE.__init__ (mae.py:41-45); the decoder
    Transformer's parameters are the 'transformer.*' entries of an ordinary ViT spec with dim=decoder_dim."""
    ph, pw = enc_cfg["patch_size"]
    pd = ph * pw * enc_cfg.get("channels", 3)
    d = enc_cfg["dim"]
    spec = []
    if d != decoder_dim:                                     # mae.py:41: Dense, else Identity
        spec += [("enc_to_dec.kernel", (d, decoder_dim)), ("enc_to_dec.bias", (decoder_dim,))]
    spec += [("mask_token", (decoder_dim,)),                 # mae.py:42
             ("decoder_pos_emb.embeddings", (num_patches, decoder_dim)),   # mae.py:44
             ("to_pixels.kernel", (decoder_dim, pd)), ("to_pixels.bias", (pd,))]   # mae.py:45
    return spec


def simmim_param_spec(enc_cfg: dict):
    ph, pw = enc_cfg["patch_size"]
    pd = ph * pw * enc_cfg.get("channels", 3)
    d = enc_cfg["dim"]
    return [("mask_token", (d,)), ("to_pixels.kernel", (d, pd)), ("to_pixels.bias", (pd,))]   # simmim.py:83-84


def _patch_tokens(enc_cfg, E, img, q):
    ph, pw = enc_cfg["patch_size"]
    patches = R.patch_unfold(img, ph, pw)                                    # to_patch          mae.py:49 / simmim.py:88
    tokens = R._dense(q(patches), E, "patch_embedding", q)                   # patch_to_emb      mae.py:53 / simmim.py:98
    n = patches.shape[1]
    pos = E["pos_embedding"][:, 1:n + 1]                                     # mae.py:54 / simmim.py:95
    return patches, tokens + pos, pos


def mae_forward(enc_cfg: dict, dec_cfg: dict, E: Dict[str, torch.Tensor], D: Dict[str, torch.Tensor], Wp: Dict[str, torch.Tensor],
                img: torch.Tensor, rand_indices: np.ndarray, masking_ratio: float, literal_loss: bool = True,
                detach_like_reference: bool = False, q=None):
    """Returns (loss, pred_pixel_values, masked_patches).  E: encoder params, D: decoder ViT-spec params (only
    'transformer.*' is used), Wp: wrapper params (mae_param_spec)."""
    q = q or R._ident
    patches, tokens, _ = _patch_tokens(enc_cfg, E, img, q)
    b, n, _ = patches.shape
    nm = num_masked(masking_ratio, n)                                        # mae.py:57
    idx = torch.as_tensor(np.ascontiguousarray(rand_indices), dtype=torch.long)
    masked, unmasked = idx[:, :nm], idx[:, nm:]                              # mae.py:59
    br = torch.arange(b)[:, None]
    tokens = tokens[br, unmasked]                                            # mae.py:62
    masked_patches = patches[br, masked]                                     # mae.py:65
    if detach_like_reference:
        tokens = tokens.detach()
    encoded = R._transformer(tokens, E, enc_cfg, "transformer", enc_cfg["depth"], q)          # mae.py:69
    if "enc_to_dec.kernel" in Wp:
        dec_tokens = R._dense(encoded, Wp, "enc_to_dec", q)                                 # mae.py:72
    else:
        dec_tokens = encoded                                                                # Identity mae.py:10-15
    dpos = Wp["decoder_pos_emb.embeddings"]
    dec_tokens = dec_tokens + dpos[unmasked]                                 # mae.py:75
    mask_tokens = Wp["mask_token"].expand(b, nm, -1) + dpos[masked]          # mae.py:78-79
    dec_in = torch.cat([mask_tokens, dec_tokens], dim=1)                     # mae.py:82
    decoded = R._transformer(dec_in, D, dec_cfg, "transformer", dec_cfg["depth"], q)          # mae.py:83
    pred = R._dense(decoded[:, :nm], Wp, "to_pixels", q)                     # mae.py:86-87
    if literal_loss:
        loss = (pred ** 2).mean()                                            # mae.py:90 as written
    else:
        loss = ((pred - masked_patches) ** 2).mean()
    return loss, pred, masked_patches


def simmim_forward(enc_cfg: dict, E: Dict[str, torch.Tensor], Wp: Dict[str, torch.Tensor], img: torch.Tensor,
                   masked_indices: np.ndarray, masking_ratio: float, detach_like_reference: bool = False, q=None):
    """Returns (loss, pred_pixel_values, masked_patches)."""
    q = q or R._ident
    patches, tokens, pos = _patch_tokens(enc_cfg, E, img, q)
    b, n, _ = patches.shape
    nm = num_masked(masking_ratio, n)                                        # simmim.py:106
    idx = torch.as_tensor(np.ascontiguousarray(masked_indices), dtype=torch.long)
    assert idx.shape == (b, nm)
    mask_tokens = Wp["mask_token"].expand(b, n, -1) + pos                    # simmim.py:102-103
    mask = torch.zeros(b, n, dtype=torch.bool)
    mask[torch.arange(b)[:, None], idx] = True                               # scatter_numpy(zeros, -1, indices, 1)  simmim.py:109-110
    tokens = torch.where(mask[..., None], mask_tokens, tokens)               # simmim.py:113
    encoded = R._transformer(tokens, E, enc_cfg, "transformer", enc_cfg["depth"], q)          # simmim.py:116
    br = torch.arange(b)[:, None]
    enc_m = encoded[br, idx]                                                 # simmim.py:119
    if detach_like_reference:
        enc_m = enc_m.detach()
    pred = R._dense(enc_m, Wp, "to_pixels", q)                               # simmim.py:122
    masked_patches = patches[br, idx]                                        # simmim.py:125
    loss = (pred - masked_patches).abs().mean() / nm                         # simmim.py:128
    return loss, pred, masked_patches


def _grads(P):
    return {k: (v.grad.detach().numpy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in P.items()}


def mae_forward_backward(enc_cfg, dec_cfg, enc_params, dec_params, wrap_params, img, rand_indices, masking_ratio,
                         literal_loss=True, detach_like_reference=False, dtype=torch.float64, q=None):
    """Returns (loss, pred, {encoder grads}, {decoder grads}, {wrapper grads}) for d(loss) = 1."""
    E = R.to_torch(enc_params, dtype, True)
    D = R.to_torch(dec_params, dtype, True)
    Wp = R.to_torch(wrap_params, dtype, True)
    loss, pred, _ = mae_forward(enc_cfg, dec_cfg, E, D, Wp, torch.tensor(img, dtype=dtype), rand_indices, masking_ratio,
                                literal_loss, detach_like_reference, q)
    loss.backward()
    return float(loss.detach()), pred.detach().numpy(), _grads(E), _grads(D), _grads(Wp)


def simmim_forward_backward(enc_cfg, enc_params, wrap_params, img, masked_indices, masking_ratio, detach_like_reference=False,
                            dtype=torch.float64, q=None):
    E = R.to_torch(enc_params, dtype, True)
    Wp = R.to_torch(wrap_params, dtype, True)
    loss, pred, _ = simmim_forward(enc_cfg, E, Wp, torch.tensor(img, dtype=dtype), masked_indices, masking_ratio,
                                   detach_like_reference, q)
    loss.backward()
    return float(loss.detach()), pred.detach().numpy(), _grads(E), _grads(Wp)


# ------------------------------------------------------------------------------------------------ MPP (mpp.py:90-218)
def mpp_num_masked(mask_prob: float, num_patches: int) -> int:
    import math
    return min(num_patches, math.ceil(mask_prob * num_patches))        # mpp.py:80


def mpp_param_spec(enc_cfg: dict, output_channel_bits: int = 3):
    """(name, shape) of MPP's own parameters in the attribute order of MPP.__init__ (mpp.py:149,159)."""
    ph, pw = enc_cfg["patch_size"]
    c = enc_cfg.get("channels", 3)
    return [("to_bits.kernel", (enc_cfg["dim"], 2 ** (output_channel_bits * c))), ("to_bits.bias", (2 ** (output_channel_bits * c),)),
            ("mask_token", (c * ph * pw,))]


def mpp_labels(img: torch.Tensor, patch: int, bits: int, max_pixel_val: float, mean=None, std=None) -> torch.Tensor:
    """MPPLoss target labels [b, num_patches] as the code evidently means them (mpp.py:104-123 with the clamp to [0, max_pixel_val])."""
    t = img
    if mean and std:
        t = t * torch.tensor(std, dtype=t.dtype) + torch.tensor(mean, dtype=t.dtype)     # mpp.py:108-109 (per channel, NHWC)
    t = t.clamp(0.0, max_pixel_val)
    b, H, W, c = t.shape
    avg = t.reshape(b, H // patch, patch, W // patch, patch, c).mean(dim=(2, 4)).reshape(b, -1, c)         # mpp.py:113
    bin_size = max_pixel_val / (2 ** bits)
    bounds = torch.tensor(np.arange(bin_size, max_pixel_val, bin_size), dtype=torch.float64)               # mpp.py:115
    disc = torch.bucketize(avg.detach().to(torch.float64), bounds, right=True)                             # Bucketize: boundaries <= value
    weights = torch.tensor([(2 ** bits) ** i for i in range(c)])                                           # mpp.py:118
    return (disc * weights).sum(-1)                                                                        # mpp.py:121


def mpp_forward(enc_cfg: dict, E: Dict[str, torch.Tensor], Wp: Dict[str, torch.Tensor], img: torch.Tensor, masked_indices: np.ndarray,
                output_channel_bits: int = 3, max_pixel_val: float = 1.0, mean=None, std=None, literal: bool = True, q=None):
    """MPP.call (mpp.py:166-218) -> (loss, logits of the masked positions [b, nm, 2^(bits c)]).  `masked_indices` int [b, nm] is the top_k draw
    of get_mask_subset_with_prob (:79-88).  literal=True restates the code AS IT RUNS under TensorFlow:
      * the replacements of :177-190 are written into `.numpy()` copies and never reach masked_input -- the transformer sees the patches;
      * the loss is tf.nn.softmax_cross_entropy_with_logits(labels=predictions, logits=label ids [n, 1]) (argument order of :125): the
        broadcast logits are all equal, log_softmax = -log(nb), loss = log(nb) * mean_i sum_j predictions_ij.
    literal=False: cross-entropy of the masked positions' logits against mpp_labels()."""
    q = q or R._ident
    ph, pw = enc_cfg["patch_size"]
    patches = R.patch_unfold(img, ph, pw)                                                   # mpp.py:175
    x = R._dense(q(patches), E, "patch_embedding", q)                                       # mpp.py:200
    b, n, _ = x.shape
    x = torch.cat([E["cls_token"].expand(b, -1, -1), x], dim=1) + E["pos_embedding"][:, :n + 1]   # mpp.py:203-208 (dropout rate 0 here)
    enc = R._transformer(x, E, enc_cfg, "transformer", enc_cfg["depth"], q)                 # mpp.py:212
    idx = torch.as_tensor(np.ascontiguousarray(masked_indices), dtype=torch.long)
    rows = enc[:, 1:][torch.arange(b)[:, None], idx]                                        # logits[:, 1:][mask]  (mpp.py:214, :125), row order irrelevant to the mean
    logits = R._dense(rows, Wp, "to_bits", q)                                               # mpp.py:213
    nb = logits.shape[-1]
    if literal:
        loss = float(np.log(nb)) * logits.sum(-1).mean()
    else:
        lab = mpp_labels(img, ph, output_channel_bits, max_pixel_val, mean, std)[torch.arange(b)[:, None], idx]
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, nb), lab.reshape(-1))
    return loss, logits
