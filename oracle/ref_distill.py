"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Pinned by tests/golden/ref_distill_*.npz (the reference's own distill.py run under oracle/tf_shim;
tests/test_ref_fixtures.py).

Torch-CPU restatement (fp64, autograd) of vit_tensorflow/distill.py: DistillMixin.call (distill.py:16-44) and
DistillWrapper.call (distill.py:107-134), with the Keras loss functions it calls written out:

  * keras.losses.categorical_crossentropy(y_true, y_pred, from_logits=True) = -sum_c y_true * log_softmax(y_pred)    [per sample]
  * keras.losses.KLDivergence(reduction=NONE)(y_true, y_pred): y_true, y_pred are clipped to [epsilon, 1] (epsilon = 1e-7),
    then sum_c y_true * log(y_true / y_pred).  distill.py:122-124 passes LOG-probabilities as y_pred; they are <= 0, so the clip
    turns every one of them into 1e-7: `literal_loss=True` restates exactly that (the term no longer depends on the student),
    `literal_loss=False` is the KL divergence the code evidently means.
  * hard=True (distill.py:130-132) hands rank-1 integer labels to categorical_crossentropy, which TensorFlow rejects with a shape
    error; restated in its sparse form (cross-entropy against the teacher's argmax), the only reading that runs.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ref_torch as R

KERAS_EPS = 1e-7


def student_forward(cfg, P, img, distill_token=None, q=None):
    """DistillMixin.call (distill.py:16-44) for DistillableViT (_attend = dropout + transformer, distill.py:54-57; dropout 0 here)."""
    q = q or R._ident
    ph, pw = cfg["patch_size"]
    x = R._dense(q(R.patch_unfold(img, ph, pw)), P, "patch_embedding", q)            # distill.py:18
    b, n, d = x.shape
    cls = P["cls_token"].expand(b, 1, d)                                             # distill.py:21
    x = torch.cat([cls, x], dim=1) + P["pos_embedding"][:, :n + 1]                   # distill.py:22-23
    distilling = distill_token is not None
    if distilling:
        x = torch.cat([x, distill_token.reshape(1, 1, d).expand(b, 1, d)], dim=1)    # distill.py:25-27
    x = R._transformer(x, P, cfg, "transformer", cfg["depth"], q)                    # distill.py:29
    dtok = None
    if distilling:
        x, dtok = x[:, :-1], x[:, -1]                                                # distill.py:32
    x = x.mean(dim=1) if cfg["pool"] == "mean" else x[:, 0]                          # distill.py:34-37
    logits = R._dense(q(R.layer_norm(x, P["mlp_head.norm.gamma"], P["mlp_head.norm.beta"])), P, "mlp_head", q)   # distill.py:39
    return (logits, dtok) if distilling else logits


def wrapper_param_spec(dim, num_classes):
    """Attribute order of DistillWrapper.__init__ (distill.py:101-106)."""
    return [("distillation_token", (1, 1, dim)), ("distill_mlp.norm.gamma", (dim,)), ("distill_mlp.norm.beta", (dim,)),
            ("distill_mlp.kernel", (dim, num_classes)), ("distill_mlp.bias", (num_classes,))]


def wrapper_loss(cfg, P, Wd, img, labels, teacher_logits, temperature=1.0, alpha=0.5, hard=False, literal_loss=True, q=None, student_fn=None):
    """DistillWrapper.call (distill.py:107-134): returns (loss [b], student_logits, distill_logits).  student_fn(cfg, P, img, token)
    replaces the DistillableViT forward (e.g. oracle.ref_t2t.student_forward for a DistillableT2TViT student, distill.py:60-72)."""
    T = temperature
    teacher_logits = teacher_logits.detach()                                                        # distill.py:114
    if student_fn is not None:
        student_logits, dtok = student_fn(cfg, P, img, Wd["distillation_token"])
    else:
        student_logits, dtok = student_forward(cfg, P, img, Wd["distillation_token"], q)            # distill.py:116
    yh = R.layer_norm(dtok, Wd["distill_mlp.norm.gamma"], Wd["distill_mlp.norm.beta"])
    distill_logits = yh @ Wd["distill_mlp.kernel"] + Wd["distill_mlp.bias"]                         # distill.py:117
    loss = -(labels * torch.log_softmax(student_logits, dim=-1)).sum(dim=-1)                        # distill.py:119
    if not hard:
        x = torch.log_softmax(distill_logits / T, dim=-1)                                           # distill.py:122
        y = torch.softmax(teacher_logits / T, dim=-1)                                               # distill.py:123
        if literal_loss:
            yt, yp = torch.clamp(y, KERAS_EPS, 1.0), torch.clamp(x, KERAS_EPS, 1.0)                 # KLDivergence's clips
            kl = (yt * torch.log(yt / yp)).sum(dim=-1)                                              # distill.py:124
        else:
            kl = (y * (torch.log(y) - x)).sum(dim=-1)
        distill_loss = kl.sum() / kl.shape[0] * T ** 2                                              # distill.py:126-129
    else:
        tl = teacher_logits.argmax(dim=-1)                                                          # distill.py:131
        distill_loss = -torch.log_softmax(distill_logits, dim=-1)[torch.arange(tl.shape[0]), tl]    # distill.py:132 (sparse form)
    return loss * (1 - alpha) + distill_loss * alpha, student_logits, distill_logits               # distill.py:134


def wrapper_forward_backward(cfg, params, wrap_params, img, labels, teacher_logits, dloss=None, dtype=torch.float64, **kw):
    P = R.to_torch(params, dtype, True)
    Wd = R.to_torch(wrap_params, dtype, True)
    loss, sl, dl = wrapper_loss(cfg, P, Wd, torch.tensor(img, dtype=dtype), torch.tensor(labels, dtype=dtype),
                                torch.tensor(teacher_logits, dtype=dtype), **kw)
    cot = torch.ones_like(loss) if dloss is None else torch.tensor(dloss, dtype=dtype)
    loss.backward(cot)
    g = lambda D: {k: (v.grad.detach().numpy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in D.items()}
    return loss.detach().numpy(), sl.detach().numpy(), dl.detach().numpy(), g(P), g(Wd)
