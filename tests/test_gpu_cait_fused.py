"""GPU tier: the one-kernel CaiT talking-heads attention forward (attn_cait_fused.hip; cait.py:109-129) against (a) the oracle (exact
restatement of cait.py, pinned by tests/golden/ref_cait_*.npz) and (b) the launch-per-op path it replaces (VITX_CAIT_FUSED=0: batched QK^T
GEMM -> mix / softmax / mix chain kernel -> batched A V GEMM) on the same weights and inputs.  Shapes cover every head-count instance
(4, 8, 12, 16), patch counts that are not multiples of the 16-row MFMA tile (36, 49) or of the 4-float score pitch (49), the BASELINE.json
cfg5 geometry (64 patches, 16 heads) and the 80-key limit (5 x 16).  The class-attention stage (1 query, n + 1 keys) stays on the
launch-per-op path in both runs."""
import numpy as np
import pytest

from oracle import ref_torch, spec
from util import rel_max_err

pytestmark = pytest.mark.gpu

CASES = {
    # name: (kwargs, batch)
    "h4_n16": (dict(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, cls_depth=1, heads=4, mlp_dim=256, dim_head=64), 3),
    "h12_n36": (dict(image_size=96, patch_size=16, num_classes=10, dim=192, depth=2, cls_depth=1, heads=12, mlp_dim=256, dim_head=64), 2),
    "h8_n49": (dict(image_size=112, patch_size=16, num_classes=10, dim=128, depth=1, cls_depth=2, heads=8, mlp_dim=256, dim_head=64), 2),
    "h16_n64": (dict(image_size=128, patch_size=16, num_classes=10, dim=256, depth=2, cls_depth=2, heads=16, mlp_dim=512, dim_head=64), 5),
    "h16_n80": (dict(image_size=(128, 160), patch_size=16, num_classes=10, dim=256, depth=1, cls_depth=1, heads=16, mlp_dim=512, dim_head=64), 2),
}
# bf16 mode against the EXACT oracle: the tolerances of the other bf16 parity tests (tests/test_gpu_parity.py)
LOGIT_TOL, GRAD_RTOL = 3.4e-2, 6.0e-2
# fused against launch-per-op: same rounding points (bf16 q/k/v, fp32 scores and mixes, bf16 mixed softmax into A V), different exp
FUSED_VS_UNFUSED_LOGIT, FUSED_VS_UNFUSED_GRAD = 1e-2, 1.5e-2


def _run(kw, b, fused, monkeypatch):
    from vit_tensorflow.cait import CaiT
    monkeypatch.setenv("VITX_CAIT_FUSED", "1" if fused else "0")
    cfg = spec.make_config("cait", **kw)
    P = spec.init_params(cfg, 1, randomize_all=True)
    m = CaiT(**kw, compute="bf16", max_batch=b, seed=0)
    m.load_state_dict({k: v.astype(np.float32) for k, v in P.items()})
    rng = np.random.Generator(np.random.PCG64(3))
    hw = kw["image_size"] if isinstance(kw["image_size"], tuple) else (kw["image_size"], kw["image_size"])
    img = rng.standard_normal((b, hw[0], hw[1], 3)).astype(np.float32)
    dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
    logits = np.asarray(m(img, training=True))
    grads, _ = m.backward(dl)
    return cfg, P, img, dl, logits, grads


@pytest.mark.parametrize("case", list(CASES))
def test_fused_talking_heads_matches_oracle_and_unfused_path(case, monkeypatch):
    kw, b = CASES[case]
    cfg, P, img, dl, lg_f, g_f = _run(kw, b, True, monkeypatch)
    _, _, _, _, lg_u, g_u = _run(kw, b, False, monkeypatch)
    ref_logits, ref_g, _ = ref_torch.forward_backward(cfg, P, img, dl)
    e_log = np.abs(lg_f - ref_logits).max() / max(1.0, ref_logits.std())
    worst = max(((rel_max_err(g_f[k], ref_g[k]), k) for k in ref_g))
    d_log = np.abs(lg_f - lg_u).max()
    d_g = max(((rel_max_err(g_f[k], np.asarray(g_u[k], np.float64)), k) for k in g_u))
    print(f"[{case}] fused vs oracle: logits {e_log:.3e}, worst grad {worst[0]:.3e} at {worst[1]}; fused vs unfused: logits {d_log:.3e}, "
          f"worst grad {d_g[0]:.3e} at {d_g[1]}")
    assert e_log <= LOGIT_TOL
    assert worst[0] <= GRAD_RTOL, worst
    assert d_log <= FUSED_VS_UNFUSED_LOGIT
    assert d_g[0] <= FUSED_VS_UNFUSED_GRAD, d_g


def test_inference_forward_keeps_nothing_and_gives_the_same_logits(monkeypatch):
    """training=False: no score tensor leaves the chip (the kept-tensor descriptors are empty); same logits as the training forward."""
    from vit_tensorflow.cait import CaiT
    monkeypatch.setenv("VITX_CAIT_FUSED", "1")
    kw, b = CASES["h16_n64"]
    m = CaiT(**kw, compute="bf16", max_batch=b, seed=0)
    rng = np.random.Generator(np.random.PCG64(5))
    img = rng.standard_normal((b, kw["image_size"], kw["image_size"], 3)).astype(np.float32)
    a = np.asarray(m(img, training=False))
    t = np.asarray(m(img, training=True))
    assert np.array_equal(a, t)


def test_fused_kernel_is_the_one_that_runs(monkeypatch):
    """The profile of one forward + backward names the fused kernel once per patch-stage block; the chain kernel's forward only runs in the
    class-attention blocks."""
    import ctypes as C
    from vit_tensorflow import _native as N
    from vit_tensorflow.cait import CaiT
    monkeypatch.setenv("VITX_CAIT_FUSED", "1")
    kw, b = CASES["h16_n64"]
    m = CaiT(**kw, compute="bf16", max_batch=b, seed=0)
    rng = np.random.Generator(np.random.PCG64(3))
    img = rng.standard_normal((b, kw["image_size"], kw["image_size"], 3)).astype(np.float32)
    m(img, training=True)
    lib, h = N.lib(), m._handle
    N.check(lib.vitx_profile_begin(h))
    m(img, training=True)
    stats = (N.KernelStat * 64)()
    ns = C.c_int32()
    N.check(lib.vitx_profile_end(h, stats, 64, C.byref(ns)))
    names = {stats[i].name.decode(): stats[i].launches for i in range(ns.value)}
    assert names.get("attn_cait_fused_fwd") == kw["depth"], names
    assert names.get("attn_headchain", 0) + names.get("attn_generic_headops", 0) <= kw["cls_depth"], names


@pytest.mark.parametrize("case", ["h4_n16", "h8_n49", "h16_n64"])
@pytest.mark.parametrize("layer_dropout", [0.0, 0.4])
def test_layerscale_vjp_on_the_layernorm_pass_equals_the_separate_pass(case, layer_dropout, monkeypatch):
    """The LayerScale VJP of a branch (cait.py:47-48: d scale = sum g f, d f = g scale, and the bias gradient of the Dense in front of it) rides on
    the LayerNorm VJP that produces g (VITX_LN_SCALE_FUSED=1, the default) or runs as a pass of its own (0).  Same d f bits (g scale rounded to bf16
    either way); the two column sums are taken in another order, and the bias gradient from g scale in fp32 instead of from its bf16 rounding.
    With layer dropout (cait.py:17-31) the branch that follows a LayerNorm VJP is the next KEPT layer's."""
    from vit_tensorflow.cait import CaiT
    kw, b = CASES[case]
    kw = dict(kw, depth=3, layer_dropout=layer_dropout)
    cfg = spec.make_config("cait", **{k: v for k, v in kw.items() if k != "layer_dropout"})
    P = spec.init_params(cfg, 1, randomize_all=True)
    rng = np.random.Generator(np.random.PCG64(11))
    img = rng.standard_normal((b, kw["image_size"], kw["image_size"], 3)).astype(np.float32)
    dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
    out = []
    for fused in ("1", "0"):
        monkeypatch.setenv("VITX_LN_SCALE_FUSED", fused)
        m = CaiT(**kw, compute="bf16", max_batch=b, seed=5)
        m.load_state_dict({k: v.astype(np.float32) for k, v in P.items()})
        logits = np.asarray(m(img, training=True, seed=1234))     # same seed: the same layers are dropped in both runs
        grads, _ = m.backward(dl)
        out.append((logits, grads))
    assert np.array_equal(out[0][0], out[1][0])
    worst = max(((rel_max_err(out[0][1][k], np.asarray(out[1][1][k], np.float64)), k) for k in out[1][1]))
    print(f"[{case}, layer dropout {layer_dropout}] LayerScale VJP fused into the LayerNorm VJP vs separate: worst grad {worst[0]:.3e} at {worst[1]}")
    assert worst[0] <= 8e-3, worst     # observed 3.5e-3 (an fc2 bias gradient: fp32 column sums against sums of bf16-rounded addends)


@pytest.mark.parametrize("variant, case", [("cait", "h16_n64"), ("cait", "h8_n49"), ("deepvit", "h16_n65"), ("deepvit", "h12_n37")])
def test_bf16_score_planes_give_the_same_bits_as_fp32_planes(variant, case, monkeypatch):
    """The mixed / normalised scores the one-kernel forwards keep for the dV product, and d(dots) on its way to the dQ / dK products, are stored as
    bf16 (VITX_SCORE_BF16=1, the default) or as fp32 (0).  The batched products round their score operand to bf16 while staging it, with the same
    round-to-nearest-even the kernels apply when they store bf16 themselves: every gradient must come out bit-identical."""
    import test_gpu_deepvit_fused as dvt
    if variant == "cait":
        from vit_tensorflow.cait import CaiT as Model
        kw, b = CASES[case]
    else:
        from vit_tensorflow.deepvit import DeepViT as Model
        kw, b = dvt.CASES[case]
    cfg = spec.make_config(variant, **kw)
    P = spec.init_params(cfg, 1, randomize_all=True)
    rng = np.random.Generator(np.random.PCG64(17))
    img = rng.standard_normal((b, kw["image_size"], kw["image_size"], 3)).astype(np.float32)
    dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
    out = []
    for lp in ("1", "0"):
        monkeypatch.setenv("VITX_SCORE_BF16", lp)
        m = Model(**kw, compute="bf16", max_batch=b, seed=5)
        m.load_state_dict({k: v.astype(np.float32) for k, v in P.items()})
        logits = np.asarray(m(img, training=True))
        grads, _ = m.backward(dl)
        out.append((logits, grads))
    assert np.array_equal(out[0][0], out[1][0])
    for k in out[0][1]:
        assert np.array_equal(out[0][1][k], out[1][1][k]), k


@pytest.mark.parametrize("case", ["h4_n16", "h16_n64"])
def test_q_and_kv_projections_as_one_dense_equal_two(case, monkeypatch):
    """Patch stage (cait.py:109-116 with context = None): to_q and to_kv read the same tokens and run as ONE Dense on concatenated bf16 operand copies
    (VITX_CAIT_QKV_CAT=1, the default) or as two (0).  Same dot products forward (bit-identical logits); backward, d(y1) comes out of one GEMM over
    d(q | k | v) instead of two GEMMs and a bf16 add (one rounding less).  Parameters and gradients stay two tensors either way."""
    from vit_tensorflow.cait import CaiT
    kw, b = CASES[case]
    cfg = spec.make_config("cait", **kw)
    P = spec.init_params(cfg, 1, randomize_all=True)
    rng = np.random.Generator(np.random.PCG64(23))
    img = rng.standard_normal((b, kw["image_size"], kw["image_size"], 3)).astype(np.float32)
    dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
    out = []
    for cat in ("1", "0"):
        monkeypatch.setenv("VITX_CAIT_QKV_CAT", cat)
        m = CaiT(**kw, compute="bf16", max_batch=b, seed=5)
        m.load_state_dict({k: v.astype(np.float32) for k, v in P.items()})
        logits = np.asarray(m(img, training=True))
        grads, _ = m.backward(dl)
        out.append((logits, grads))
    assert np.array_equal(out[0][0], out[1][0])
    assert set(out[0][1]) == set(out[1][1])
    worst = max(((rel_max_err(out[0][1][k], np.asarray(out[1][1][k], np.float64)), k) for k in out[1][1]))
    print(f"[{case}] q / kv as one Dense vs two: worst grad {worst[0]:.3e} at {worst[1]}")
    assert worst[0] <= 8e-3, worst


@pytest.mark.parametrize("case", ["h4_n16", "h16_n64"])
def test_second_backward_on_the_same_forward_gives_the_same_gradients(case, monkeypatch):
    """ The bf16 talking-heads backward writes d(dots) over the forward's kept mixed softmax.  A second backward on the same forward is
    allowed by the API: it must notice that the kept tensor is gone and recompute the scores, not read d(dots) as attention weights."""
    from vit_tensorflow.cait import CaiT
    monkeypatch.setenv("VITX_CAIT_FUSED", "1")
    kw, b = CASES[case]
    m = CaiT(**kw, compute="bf16", max_batch=b, seed=0)
    rng = np.random.Generator(np.random.PCG64(11))
    img = rng.standard_normal((b, kw["image_size"], kw["image_size"], 3)).astype(np.float32)
    dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
    m(img, training=True)
    g1, _ = m.backward(dl)
    g1 = {k: np.array(v, copy=True) for k, v in g1.items()}
    g2, _ = m.backward(dl)
    worst = max(((rel_max_err(g2[k], np.asarray(g1[k], np.float64)), k) for k in g1))
    print(f"[{case}] second backward vs first: worst {worst[0]:.3e} at {worst[1]}")
    assert worst[0] <= 1e-5, worst   # recomputed scores have the forward's bits; only reduction order could differ
