"""SURVEY.md section 8 "next" row f4, the last two siblings: the efficient.ViT shell around a caller-supplied transformer
(efficient.py:12-56) and the T2T tokenizer tf.image.extract_patches(..., 'SAME') (t2t.py:39-47).
CPU tier: the oracle's loop form against an independent formulation, TensorFlow's SAME geometry through the C ABI, finite
differences of the VJP.  GPU tier: kernels bit-exact against the oracle; shell logits / gradients against the oracle in both
compute modes with an engine transformer, a torch module and a token-count-changing callable in the middle."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ref_efficient as R, ref_torch, spec
from vit_tensorflow import _native as N

GEOMS = [  # (H, W, C, k, stride)
    (224, 224, 3, 7, 4), (56, 56, 5, 3, 2), (28, 28, 4, 3, 2),      # T2TViT's default t2t_layers (t2t.py:52) on 224 px
    (9, 7, 2, 3, 2), (8, 8, 1, 3, 1), (5, 5, 3, 7, 4), (6, 10, 2, 2, 3), (4, 4, 3, 1, 1), (7, 7, 1, 4, 2),
]


def test_oracle_loop_form_matches_pad_plus_unfold():
    rng = np.random.default_rng(0)
    for H, W, Cc, k, s in GEOMS[3:]:
        x = rng.standard_normal((2, H, W, Cc))
        assert np.array_equal(R.extract_patches(x, k, s), R.extract_patches_unfold(torch.tensor(x), k, s).numpy()), (H, W, Cc, k, s)
    # a hand-checked case: 1x4x4x1 image 0..15, k=3, s=2 -> out 2x2, pad_total = 1 -> nothing before, one row / column after
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    y = R.extract_patches(x, 3, 2)
    assert y.shape == (1, 2, 2, 9)
    assert y[0, 0, 0].tolist() == [0, 1, 2, 4, 5, 6, 8, 9, 10] and y[0, 1, 1].tolist() == [10, 11, 0, 14, 15, 0, 0, 0, 0]


def test_same_geometry_through_the_c_abi_and_t2t_sizes():
    from vit_tensorflow import t2t
    for H, W, Cc, k, s in GEOMS:
        oh, ow, f = t2t.extract_patches_shape(H, W, Cc, k, s)
        assert (oh, ow, f) == (R.same_padding(H, k, s)[0], R.same_padding(W, k, s)[0], k * k * Cc)
    # t2t.py:62-66: the sizes T2TViT derives with conv_output_size(size, k, stride, stride // 2) are what SAME produces at 224 px
    size = 224
    for k, s in ((7, 4), (3, 2), (3, 2)):
        nxt = t2t.conv_output_size(size, k, s, s // 2)
        assert nxt == t2t.extract_patches_shape(size, size, 3, k, s)[0]
        size = nxt
    assert size == 14
    lib = N.lib()
    assert lib.vitx_extract_patches_shape(8, 8, 3, 0, 1, None, None, None) == N.ERR_INVALID
    buf = np.zeros(4, np.float32)
    assert lib.vitx_extract_patches(None, 1, 2, 2, 1, 1, 1, buf.ctypes.data_as(C.c_void_p)) == N.ERR_INVALID
    assert lib.vitx_extract_patches(buf.ctypes.data_as(C.c_void_p), 0, 2, 2, 1, 1, 1, buf.ctypes.data_as(C.c_void_p)) == N.OK   # empty batch


def test_efficient_shell_oracle_constructor_and_table():
    from vit_tensorflow.efficient import ViT
    with pytest.raises(AssertionError, match='image dimensions must be divisible by the patch size'):      # efficient.py:18
        ViT(image_size=30, patch_size=8, num_classes=5, dim=32, transformer=None)
    with pytest.raises(AssertionError, match='pool type must be either cls'):                               # efficient.py:19
        ViT(image_size=32, patch_size=8, num_classes=5, dim=32, transformer=None, pool='max')
    m = ViT(image_size=(32, 16), patch_size=8, num_classes=5, dim=32, transformer=lambda x, training=True: x)
    cfg = spec.make_config("vit", image_size=(32, 16), patch_size=8, num_classes=5, dim=32, depth=0, heads=1, mlp_dim=64, dim_head=64)
    assert [(a, tuple(s)) for a, s, _ in m._table] == [(a, tuple(s)) for a, s, _ in spec.param_spec(cfg)]
    assert [a for a, _, _ in m._table] == ["pos_embedding", "cls_token", "patch_embedding.kernel", "patch_embedding.bias",
                                           "mlp_head.norm.gamma", "mlp_head.norm.beta", "mlp_head.kernel", "mlp_head.bias"]
    # the oracle shell with the identity in the middle is the depth-0 ViT of ref_torch
    P = spec.init_params(cfg, 3, randomize_all=True)
    rng = np.random.default_rng(1)
    img, dl = rng.standard_normal((2, 32, 16, 3)), rng.standard_normal((2, 5))
    l0, g0, _ = ref_torch.forward_backward(cfg, P, img, dl)
    l1, g1, _, tok = R.shell_forward_backward(cfg, P, img, dl, lambda t: t)
    assert np.abs(l0 - l1).max() < 1e-12 and all(np.abs(g0[k] - g1[k]).max() < 1e-12 for k in g0) and tok.shape == (2, 9, 32)


def _fd_check(f, x, dx_analytic, rng, n=4, eps=1e-6):
    for _ in range(n):
        dirn = rng.standard_normal(x.shape)
        fd = (f(x + eps * dirn) - f(x - eps * dirn)) / (2 * eps)
        assert abs(fd - float((dx_analytic * dirn).sum())) <= 1e-6 * max(1.0, abs(fd))


def test_oracle_extract_patches_vjp_by_finite_differences():
    rng = np.random.default_rng(2)
    for H, W, Cc, k, s in GEOMS[3:6]:
        x = rng.standard_normal((2, H, W, Cc))
        w = rng.standard_normal(R.extract_patches(x, k, s).shape)
        xt = torch.tensor(x, requires_grad=True)
        (R.extract_patches_unfold(xt, k, s) * torch.tensor(w)).sum().backward()
        _fd_check(lambda a: float((R.extract_patches(a, k, s) * w).sum()), x, xt.grad.numpy(), rng)


# ------------------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize("geom", GEOMS)
def test_extract_patches_kernel_is_bit_exact(geom):
    from vit_tensorflow import t2t
    H, W, Cc, k, s = geom
    rng = np.random.default_rng(3)
    b = 2 if H * W > 10000 else 3
    x = rng.standard_normal((b, H, W, Cc)).astype(np.float32)
    ref = R.extract_patches_unfold(torch.tensor(x), k, s).numpy()
    if H * W <= 100:
        assert np.array_equal(ref, R.extract_patches(x, k, s))
    y = t2t.extract_patches(x, k, s)
    assert y.dtype == np.float32 and y.shape == ref.shape and np.array_equal(y, ref)                       # pure indexing: bit-exact
    # VJP: overlapping windows are summed in a fixed order -> compare with autograd of the independent formulation
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    xt = torch.tensor(x.astype(np.float64), requires_grad=True)
    (R.extract_patches_unfold(xt, k, s) * torch.tensor(dy.astype(np.float64))).sum().backward()
    dx = t2t.extract_patches_backward(dy, x.shape, k, s)
    assert np.abs(dx - xt.grad.numpy()).max() <= 1e-5 * max(1.0, np.abs(xt.grad.numpy()).max())
    if s >= k:   # disjoint windows: every input element has at most one tap, so the VJP is pure indexing too
        assert np.array_equal(dx, xt.grad.numpy().astype(np.float32))
    assert np.array_equal(t2t.extract_patches_backward(dy, x.shape, k, s), dx)                               # deterministic


@pytest.mark.gpu
def test_rearrange_unfold_layers_chain_like_t2t():
    from vit_tensorflow import t2t
    rng = np.random.default_rng(4)
    img = rng.standard_normal((2, 32, 32, 3)).astype(np.float32)
    l1, l2 = t2t.RearrangeUnfold(True, 7, 4), t2t.RearrangeUnfold(False, 3, 2)
    t1 = l1(img)
    t2_ = l2(t1)
    r1 = R.rearrange_unfold(img, True, 7, 4)
    r2 = R.rearrange_unfold(r1, False, 3, 2)
    assert t1.shape == (2, 64, 147) and t2_.shape == (2, 16, 1323)
    assert np.array_equal(t1, r1) and np.array_equal(t2_, r2)
    d2 = rng.standard_normal(t2_.shape).astype(np.float32)
    dimg = l1.backward(l2.backward(d2))
    xt = torch.tensor(img.astype(np.float64), requires_grad=True)
    a = R.extract_patches_unfold(xt, 7, 4).reshape(2, 8, 8, 147)
    bb = R.extract_patches_unfold(a, 3, 2).reshape(2, 16, 1323)
    (bb * torch.tensor(d2.astype(np.float64))).sum().backward()
    assert dimg.shape == img.shape and np.abs(dimg - xt.grad.numpy()).max() <= 1e-5 * np.abs(xt.grad.numpy()).max()


class _TorchBlock(torch.nn.Module):
    """A caller-supplied transformer that is not ours: pre-norm MLP block with a residual, fp32."""

    def __init__(self, dim):
        super().__init__()
        g = torch.Generator().manual_seed(7)
        self.norm = torch.nn.LayerNorm(dim)
        self.fc1, self.fc2 = torch.nn.Linear(dim, 2 * dim), torch.nn.Linear(2 * dim, dim)
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.1

    def forward(self, x):
        return x + self.fc2(torch.tanh(self.fc1(self.norm(x))))


SHELL = {"fp32": dict(image_size=(32, 48), patch_size=8, num_classes=7, dim=32),
         "bf16": dict(image_size=64, patch_size=16, num_classes=10, dim=128)}


@pytest.mark.gpu
@pytest.mark.parametrize("compute,pool,middle", [("fp32", "cls", "torch"), ("fp32", "mean", "torch"), ("fp32", "cls", "drop_tokens"),
                                                 ("fp32", "mean", "engine"), ("bf16", "cls", "torch"), ("bf16", "mean", "engine")])
def test_efficient_vit_shell_matches_the_oracle(compute, pool, middle):
    from vit_tensorflow import ViT as FullViT
    from vit_tensorflow.efficient import ViT
    kw = SHELL[compute]
    d, b = kw["dim"], 3
    cfg = spec.make_config("vit", **kw, depth=0, heads=1, mlp_dim=64, dim_head=64, pool=pool)
    P = spec.init_params(cfg, 11, randomize_all=True)
    q = ref_torch.bf16_round if compute == "bf16" else None
    tp = None
    if middle == "torch":
        blk = _TorchBlock(d)
        mid = blk
        blk64 = _TorchBlock(d).double()
        ref_mid = lambda t: blk64(t)
    elif middle == "drop_tokens":      # a callable that changes the token count (keeps cls + every other patch), with its own VJP
        class Drop:
            def __call__(self, x, training=True):
                self.n = x.shape[1]
                return np.ascontiguousarray(x[:, ::2]) * 2.0

            def backward(self, dout):
                dx = np.zeros((dout.shape[0], self.n, dout.shape[2]), np.float32)
                dx[:, ::2] = 2.0 * dout
                return dx
        mid = Drop()
        ref_mid = lambda t: t[:, ::2] * 2.0
    else:                               # the transformer of another model of this package (efficient.py's intended use)
        tkw = dict(image_size=kw["image_size"], patch_size=kw["patch_size"], num_classes=3, dim=d, depth=2, heads=2, mlp_dim=2 * d, dim_head=d // 2)
        tcfg = spec.make_config("vit", **tkw)
        tp = spec.init_params(tcfg, 12, randomize_all=True)
        donor = FullViT(**tkw, compute=compute, max_batch=b, seed=0)
        donor.load_state_dict({k: np.asarray(v, np.float32) for k, v in tp.items()})
        mid = donor.transformer
        Pt = ref_torch.to_torch(tp, requires_grad=True)
        ref_mid = lambda t: ref_torch._transformer(t, Pt, tcfg, "transformer", tcfg["depth"], q or ref_torch._ident)
    m = ViT(**kw, transformer=mid, pool=pool, compute=compute, max_batch=b, seed=0)
    m.load_state_dict({k: np.asarray(v, np.float32) for k, v in P.items()})
    rng = np.random.default_rng(13)
    H, W = spec.pair(kw["image_size"])
    img = rng.standard_normal((b, H, W, 3)).astype(np.float32)
    dl = (rng.standard_normal((b, kw["num_classes"])) / 2).astype(np.float32)
    rl, rg, rdimg, _ = R.shell_forward_backward(cfg, P, img, dl, ref_mid, q=q)
    ltol, gtol = (1e-4, 2e-4) if compute == "fp32" else (3e-2, 6e-2)      # bf16: against the oracle with the same rounding points
    for rep in range(2):
        logits = m(img, training=False)
        grads, dimg = m.backward(dl, want_dimg=True)
        assert logits.shape == rl.shape and np.abs(logits - rl).max() <= ltol * max(1.0, np.abs(rl).max())
        for k, r in rg.items():
            assert np.abs(grads[k] - r).max() <= gtol * max(1e-6, np.abs(r).max()) + 1e-7, (rep, k)
        assert np.abs(dimg - rdimg).max() <= gtol * max(1e-6, np.abs(rdimg).max()) + 1e-7
    if middle == "engine":              # the donor's transformer gradients came back through last_transformer_grads
        tg = m.last_transformer_grads
        checked = 0
        for k, v in Pt.items():
            if k.startswith("transformer.") and v.grad is not None:
                r = v.grad.numpy()
                assert np.abs(tg[k] - r).max() <= gtol * max(1e-6, np.abs(r).max()) + 1e-7, k
                checked += 1
        assert checked >= 10
    if middle == "torch":
        assert m.last_transformer_grads and all(v is not None for v in m.last_transformer_grads.values())
    # another batch size on the same objects (round 6: the bf16 operand buffers' row padding is re-established per geometry; nothing else may be wiped)
    for b2 in (1, 2):
        rl2, rg2, rdimg2, _ = R.shell_forward_backward(cfg, P, img[:b2], dl[:b2], ref_mid, q=q)
        l2 = m(img[:b2], training=False)
        g2, dimg2 = m.backward(dl[:b2], want_dimg=True)
        assert np.abs(l2 - rl2).max() <= ltol * max(1.0, np.abs(rl2).max()), b2
        for k, r in rg2.items():
            assert np.abs(g2[k] - r).max() <= gtol * max(1e-6, np.abs(r).max()) + 1e-7, (b2, k)
        assert np.abs(dimg2 - rdimg2).max() <= gtol * max(1e-6, np.abs(rdimg2).max()) + 1e-7, b2
    # smaller image than configured: pos_embedding is sliced (efficient.py:45) and the unused rows get zero gradient
    if compute == "fp32" and middle == "torch":
        img2 = img[:, :16, :24]
        rl2, rg2, _, _ = R.shell_forward_backward(cfg, P, img2, dl, ref_mid)
        l2 = m(img2, training=False)
        g2, _ = m.backward(dl)
        assert np.abs(l2 - rl2).max() <= ltol * max(1.0, np.abs(rl2).max())
        assert np.abs(g2["pos_embedding"] - rg2["pos_embedding"]).max() <= gtol * np.abs(rg2["pos_embedding"]).max() and not g2["pos_embedding"][0, 7:].any()


@pytest.mark.gpu
def test_shell_entry_points_report_call_order_and_ranges():
    from vit_tensorflow.efficient import ViT
    m = ViT(**SHELL["fp32"], transformer=lambda x, training=True: x, max_batch=2, seed=0)
    with pytest.raises(N.VitxError, match="backward requires a preceding forward"):
        m.backward(np.zeros((1, 7), np.float32))
    m.build((2,))
    lib, h = N.lib(), m._handle
    buf = np.zeros(2 * 25 * 32, np.float32)
    p = buf.ctypes.data_as(C.c_void_p)
    assert lib.vitx_head_backward(h, p, p) == N.ERR_STATE and b"head_forward" in lib.vitx_last_error()
    assert lib.vitx_embed_backward(h, p, None) == N.ERR_STATE and b"embed_forward" in lib.vitx_last_error()
    assert lib.vitx_head_forward(h, p, 3, 4, p) == N.ERR_INVALID          # b > max_batch
    assert lib.vitx_head_forward(h, p, 1, 1000, p) == N.ERR_INVALID       # more tokens than the handle holds
    with pytest.raises(N.VitxError, match="no backward"):
        m(np.zeros((1, 32, 48, 3), np.float32))
        m.backward(np.zeros((1, 7), np.float32))


# ------------------------------------------------------------------------------------------------
# the rest of the surface the reference's wrappers reach into (SURVEY.md section 8b "extended surface"):
# patch_embedding(.layers), pos_embedding[...], cls_token, dropout, transformer, pool, mlp_head
@pytest.mark.gpu
@pytest.mark.parametrize("compute,name", [("fp32", "vit_small"), ("fp32", "vit_rect_mean"), ("bf16", "vit_bf16_small"), ("fp32", "deepvit_small")])
def test_wrapper_surface_composes_to_the_forward(compute, name):
    from oracle import ref_numpy
    from util import make_engine_model, oracle_cfg, rand_images
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, 3, randomize_all=True)
    b = 3
    m = make_engine_model(name, compute, max_batch=b, params=P)
    img = rand_images(cfg, b, seed=5)
    # mae.py:36-38 / simmim.py:78-80
    num_patches, encoder_dim = m.pos_embedding.shape[-2:]
    to_patch, patch_to_emb = m.patch_embedding.layers[:2]
    ph, pw = cfg["patch_size"]
    assert encoder_dim == cfg["dim"] and patch_to_emb.weights[0].shape[0] == ph * pw * 3
    patches = to_patch(img)
    assert np.array_equal(patches, ref_numpy.patch_unfold(img, ph, pw))                                   # pure indexing: bit-exact
    tokens = patch_to_emb(patches)
    ref_tok = patches.astype(np.float64) @ P["patch_embedding.kernel"] + P["patch_embedding.bias"]
    ttol = 1e-5 if compute == "fp32" else 2e-2
    assert np.abs(tokens - ref_tok).max() <= ttol * np.abs(ref_tok).max()
    # DistillMixin.call (distill.py:19-40) written against the model's attributes, as a user of the reference would
    x = m.patch_embedding(img)
    bb, n, d = x.shape
    cls_tokens = np.repeat(np.asarray(m.cls_token), bb, axis=0)
    x = np.concatenate([cls_tokens, x], axis=1)
    x = x + m.pos_embedding[:, :(n + 1)]
    x = m.dropout(x, training=False)
    x = m.transformer(x, training=False)
    x = x.mean(axis=1) if m.pool == 'mean' else x[:, 0]
    logits = m.mlp_head(np.ascontiguousarray(x))
    rl = ref_torch.forward(cfg, ref_torch.to_torch(P), torch.tensor(img, dtype=torch.float64), ref_torch.bf16_round if compute == "bf16" else None).numpy()
    ltol = 1e-4 if compute == "fp32" else 3e-2
    assert np.abs(logits - rl).max() <= ltol * max(1.0, np.abs(rl).max())
    full = m(img, training=False)
    assert np.abs(logits - full).max() <= ltol * max(1.0, np.abs(rl).max())
    # a stand-alone head call overwrites the head state of the full forward: its backward must refuse, not return wrong gradients
    m.mlp_head(np.ascontiguousarray(x))
    with pytest.raises(N.VitxError, match="preceding forward"):
        m.backward(np.zeros((b, cfg["num_classes"]), np.float32))
    m(img, training=False)
    grads, _ = m.backward(np.ones((b, cfg["num_classes"]), np.float32))                                    # and works again after a forward
    assert np.isfinite(grads["mlp_head.kernel"]).all()
    with pytest.raises(NotImplementedError):
        type(m.dropout)(0.5)(x, training=True)


# ------------------------------------------------------------------------------------------------
# committed golden fixture (tests/golden/tokenizer_and_shell.npz, written by `python -m oracle.gen_golden`)
def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "tokenizer_and_shell.npz"))


def test_oracle_reproduces_the_golden_fixture():
    from oracle.gen_golden import SHELL_KW, TOKENIZER_GEOMS, shell_middle
    z = _golden()
    assert [tuple(int(v) for v in g) for g in z["geoms"]] == TOKENIZER_GEOMS
    for i, (H, W, Cc, k, s) in enumerate(TOKENIZER_GEOMS):
        assert np.array_equal(R.extract_patches(z[f"x{i}"], k, s), z[f"y{i}"])
        xt = torch.tensor(z[f"x{i}"].astype(np.float64), requires_grad=True)
        (R.extract_patches_unfold(xt, k, s) * torch.tensor(z[f"dy{i}"].astype(np.float64))).sum().backward()
        assert np.abs(xt.grad.numpy() - z[f"dx{i}"]).max() < 1e-12
    cfg = spec.make_config("vit", **SHELL_KW, depth=0, heads=1, mlp_dim=64, dim_head=64)
    P = spec.init_params(cfg, seed=11, randomize_all=True)
    assert abs(sum(float(np.abs(v).sum()) for v in P.values()) - float(z["shell/param_checksum"])) < 1e-9
    logits, grads, dimg, _ = R.shell_forward_backward(cfg, P, z["shell/img"], z["shell/dlogits"], shell_middle)
    assert np.abs(logits - z["shell/logits"]).max() < 1e-12 and np.abs(dimg - z["shell/dimg"]).max() < 1e-12
    for k_, v in grads.items():
        assert np.abs(v - z["shell/grad/" + k_]).max() <= 1e-6 * max(1.0, np.abs(v).max()), k_      # stored as fp32


@pytest.mark.gpu
def test_kernels_match_the_golden_fixture():
    from oracle.gen_golden import SHELL_KW, TOKENIZER_GEOMS
    from vit_tensorflow import t2t
    from vit_tensorflow.efficient import ViT
    z = _golden()
    for i, (H, W, Cc, k, s) in enumerate(TOKENIZER_GEOMS):
        assert np.array_equal(t2t.extract_patches(z[f"x{i}"], k, s), z[f"y{i}"])                            # bit-exact
        dx = t2t.extract_patches_backward(z[f"dy{i}"], z[f"x{i}"].shape, k, s)
        assert np.abs(dx - z[f"dx{i}"]).max() <= 1e-5 * max(1.0, np.abs(z[f"dx{i}"]).max())

    class Drop:      # the fixture's middle (oracle.gen_golden.shell_middle) with its VJP
        def __call__(self, x, training=True):
            self.n = x.shape[1]
            return np.ascontiguousarray(x[:, ::2]) * 2.0

        def backward(self, dout):
            dx = np.zeros((dout.shape[0], self.n, dout.shape[2]), np.float32)
            dx[:, ::2] = 2.0 * dout
            return dx
    cfg = spec.make_config("vit", **SHELL_KW, depth=0, heads=1, mlp_dim=64, dim_head=64)
    P = spec.init_params(cfg, seed=11, randomize_all=True)
    m = ViT(**SHELL_KW, transformer=Drop(), max_batch=3, seed=0)
    m.load_state_dict({k_: np.asarray(v, np.float32) for k_, v in P.items()})
    logits = m(z["shell/img"], training=False)
    grads, dimg = m.backward(z["shell/dlogits"], want_dimg=True)
    assert np.abs(logits - z["shell/logits"]).max() <= 1e-4 * max(1.0, np.abs(z["shell/logits"]).max())
    for k_ in grads:
        r = z["shell/grad/" + k_]
        assert np.abs(grads[k_] - r).max() <= 2e-4 * max(1e-6, np.abs(r).max()) + 1e-7, k_
    assert np.abs(dimg - z["shell/dimg"]).max() <= 2e-4 * np.abs(z["shell/dimg"]).max() + 1e-7
