"""GPU tier: dropout rows (a4, c4).  Parity with TF's unseeded global RNG is impossible, so the
engine's counter-based dropout is tested for what the reference's semantics require: identity at inference,
inverted scaling, keep-rate, determinism per seed, and backward using the SAME masks (checked by a directional
finite difference of the engine's own forward)."""
import numpy as np
import pytest

from oracle import ref_numpy, spec
from util import make_engine_model, oracle_cfg, rand_images

pytestmark = pytest.mark.gpu


def _model(variant="vit", **over):
    kw = dict(image_size=64, patch_size=16, num_classes=10, dim=64, depth=2, heads=2, mlp_dim=128, dim_head=32)
    kw.update(over)
    if variant == "vit":
        from vit_tensorflow import ViT
        return ViT(**kw, compute="fp32", max_batch=4, seed=3), spec.make_config("vit", **{k: v for k, v in kw.items() if k not in ("dropout", "emb_dropout")})
    from vit_tensorflow.cait import CaiT
    kw.update(cls_depth=2, depth=4)
    return CaiT(**kw, compute="fp32", max_batch=4, seed=3), spec.make_config("cait", **{k: v for k, v in kw.items() if k not in ("dropout", "emb_dropout", "layer_dropout")})


def test_dropout_is_identity_at_inference_and_active_in_training():
    m, cfg = _model(dropout=0.3, emb_dropout=0.2)
    img = rand_images(cfg, 4)
    P = m.state_dict()
    ref = ref_numpy.forward(cfg, P, img)
    assert np.abs(m(img, training=False) - ref).max() <= 1e-3          # Keras Dropout is the identity when not training
    a = m(img, training=True, seed=11)
    b = m(img, training=True, seed=11)
    c = m(img, seed=12)                                                # training=True is the reference's default (vit.py:159)
    assert np.array_equal(a, b), "same seed -> same masks"
    assert np.abs(a - c).max() > 1e-3 and np.abs(a - ref).max() > 1e-3


def test_dropout_keep_rate_and_scaling():
    """Post-GELU dropout (vit.py:41): act = gelu(hpre) * m / (1 - rate) with m ~ Bernoulli(1 - rate), on the same forward."""
    from scipy.special import erf
    rate = 0.4
    m, cfg = _model(dropout=rate, depth=1)
    img = rand_images(cfg, 4)
    m(img, training=True, seed=5)
    hpre = m.debug_read("hpre", 0).astype(np.float64)
    act = m.debug_read("act", 0)
    kept = act != 0
    frac = 1.0 - kept.mean()
    assert abs(frac - rate) < 0.03, frac
    gelu = 0.5 * hpre * (1.0 + erf(hpre / np.sqrt(2.0)))
    assert np.allclose(act[kept], gelu[kept] / (1 - rate), rtol=1e-4, atol=1e-6)   # inverted dropout
    m(img, training=False)
    assert np.allclose(m.debug_read("act", 0), 0.5 * (h := m.debug_read("hpre", 0).astype(np.float64)) * (1.0 + erf(h / np.sqrt(2.0))), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("variant,over", [("vit", dict(dropout=0.25, emb_dropout=0.25)), ("cait", dict(dropout=0.2, emb_dropout=0.1, layer_dropout=0.3))])
def test_backward_reuses_the_forward_masks(variant, over):
    m, cfg = _model(variant, **over)
    img = rand_images(cfg, 3)
    rng = np.random.default_rng(0)
    dl = rng.standard_normal((3, cfg["num_classes"])).astype(np.float32)
    seed = 21
    m(img, training=True, seed=seed)
    grads, _ = m.backward(dl)
    w0 = m.get_weights()
    names = [w.name for w in m.weights]
    v = [rng.standard_normal(a.shape).astype(np.float32) * (np.abs(a).mean() + 1e-3) for a in w0]
    analytic = sum(float((grads[n].astype(np.float64) * vi).sum()) for n, vi in zip(names, v))
    eps = 2e-3
    vals = []
    for sgn in (+1, -1):
        m.set_weights([a + sgn * eps * vi for a, vi in zip(w0, v)])
        vals.append(float((m(img, training=True, seed=seed).astype(np.float64) * dl).sum()))
    m.set_weights(w0)
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - analytic) <= 3e-2 * max(1.0, abs(analytic)), (fd, analytic)
    if variant == "cait":   # parameters of skipped blocks receive exactly zero gradient (cait.py:147)
        zero_blocks = {n.rsplit(".mlp", 1)[0] for n in names if ".mlp.fc1.kernel" in n and not np.any(grads[n])}
        live_blocks = {n.rsplit(".mlp", 1)[0] for n in names if ".mlp.fc1.kernel" in n and np.any(grads[n])}
        assert live_blocks, "at least one layer per stage survives"
        for blk in zero_blocks:
            assert not np.any(grads[blk + ".attn.to_out.kernel"])
