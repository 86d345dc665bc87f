"""CPU tier for the distillation path ("next" row f3): the oracle restatement of DistillMixin.call and
DistillWrapper.call (oracle/ref_distill.py) against the plain ViT oracle, closed forms of the Keras losses, and finite differences."""
import numpy as np
import pytest
import torch

from oracle import ref_distill as RD, ref_torch, spec

CFG = dict(image_size=32, patch_size=8, num_classes=6, dim=16, depth=2, heads=2, mlp_dim=32, dim_head=8)


def _setup(pool="cls", seed=0, b=3):
    cfg = spec.make_config("vit", **CFG, pool=pool)
    P = spec.init_params(cfg, 1 + seed, True)
    rng = np.random.default_rng(2 + seed)
    Wd = {n: rng.standard_normal(s) * 0.4 for n, s in RD.wrapper_param_spec(16, 6)}
    img = rng.standard_normal((b, 32, 32, 3))
    labels = np.eye(6)[rng.integers(0, 6, b)]
    teacher = rng.standard_normal((b, 6)) * 2
    return cfg, P, Wd, img, labels, teacher


@pytest.mark.parametrize("pool", ["cls", "mean"])
def test_student_without_token_is_the_plain_vit(pool):
    cfg, P, _, img, _, _ = _setup(pool)
    Pt = ref_torch.to_torch(P)
    a = RD.student_forward(cfg, Pt, torch.tensor(img))
    assert np.abs(a.numpy() - ref_torch.forward(cfg, Pt, torch.tensor(img)).numpy()).max() < 1e-12
    logits, dtok = RD.student_forward(cfg, Pt, torch.tensor(img), torch.tensor(np.ones(16)))
    assert logits.shape == (3, 6) and dtok.shape == (3, 16)
    assert np.abs(logits.numpy() - a.numpy()).max() > 1e-6     # the extra token is attended to (distill.py:27-29)


def test_literal_soft_term_is_constant_in_the_student():
    cfg, P, Wd, img, labels, teacher = _setup()
    loss, sl, dl, gP, gW = RD.wrapper_forward_backward(cfg, P, Wd, img, labels, teacher, temperature=2.0, alpha=0.3, literal_loss=True)
    y = np.clip(torch.softmax(torch.tensor(teacher) / 2.0, -1).numpy(), 1e-7, 1.0)
    const = (y * np.log(y / 1e-7)).sum(-1).sum() / 3 * 4.0          # Keras KLDivergence with y_pred clipped to 1e-7 (distill.py:122-129)
    ce = -(labels * torch.log_softmax(torch.tensor(sl), -1).numpy()).sum(-1)
    assert np.abs(loss - (ce * 0.7 + const * 0.3)).max() < 1e-10
    assert not gW["distill_mlp.kernel"].any() and not gW["distill_mlp.norm.gamma"].any()   # nothing reaches the distillation head
    assert gW["distillation_token"].any()                                                 # ...but the token still shapes the cls path
    assert gP["mlp_head.kernel"].any()


def test_intended_soft_term_and_hard_term():
    cfg, P, Wd, img, labels, teacher = _setup(seed=1)
    loss, sl, dl, gP, gW = RD.wrapper_forward_backward(cfg, P, Wd, img, labels, teacher, temperature=3.0, alpha=0.5, literal_loss=False)
    pt = torch.softmax(torch.tensor(teacher) / 3.0, -1).numpy()
    lq = torch.log_softmax(torch.tensor(dl) / 3.0, -1).numpy()
    kl = (pt * (np.log(pt) - lq)).sum(-1)
    assert (kl >= 0).all()
    ce = -(labels * torch.log_softmax(torch.tensor(sl), -1).numpy()).sum(-1)
    assert np.abs(loss - (ce * 0.5 + kl.sum() / 3 * 9.0 * 0.5)).max() < 1e-10
    assert gW["distill_mlp.kernel"].any()
    lh, sl, dl, _, _ = RD.wrapper_forward_backward(cfg, P, Wd, img, labels, teacher, alpha=0.25, hard=True)
    hce = -torch.log_softmax(torch.tensor(dl), -1).numpy()[np.arange(3), teacher.argmax(-1)]
    assert np.abs(lh - (ce * 0.75 + hce * 0.25)).max() < 1e-10


@pytest.mark.parametrize("kw", [dict(literal_loss=False, temperature=2.0), dict(hard=True), dict(literal_loss=True)])
def test_wrapper_gradients_against_finite_differences(kw):
    cfg, P, Wd, img, labels, teacher = _setup("mean", seed=2, b=2)
    dloss = np.array([0.7, -1.3])
    run = lambda Pp, Ww: RD.wrapper_forward_backward(cfg, Pp, Ww, img, labels, teacher, dloss=dloss, alpha=0.4, **kw)
    out = run(P, Wd)
    rng = np.random.default_rng(0)
    for group, name in (("w", "distillation_token"), ("w", "distill_mlp.kernel"), ("p", "transformer.0.attn.to_qkv.kernel"), ("p", "pos_embedding")):
        base = Wd if group == "w" else P
        g = out[4][name] if group == "w" else out[3][name]
        dirn = rng.standard_normal(base[name].shape)
        eps = 1e-5
        plus, minus = dict(base), dict(base)
        plus[name] = base[name] + eps * dirn
        minus[name] = base[name] - eps * dirn
        lp = run(P, plus)[0] if group == "w" else run(plus, Wd)[0]
        lm = run(P, minus)[0] if group == "w" else run(minus, Wd)[0]
        fd = float(((lp - lm) * dloss).sum() / (2 * eps))
        assert abs(fd - float((g * dirn).sum())) <= 2e-6 * max(1.0, abs(fd)), name
