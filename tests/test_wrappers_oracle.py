"""CPU tier for the masked-image-modelling wrappers ("next" row f2): the oracle restatement of
MAE.call / SimMIM.call (oracle/ref_wrappers.py) checked against independent index arithmetic, against the properties the
algorithm must have, and against finite differences; plus the host-side pieces of the drop-in (parameter order, index draws)."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy, ref_torch, ref_wrappers as RW, spec

ENC = dict(image_size=32, patch_size=8, num_classes=5, dim=16, depth=1, heads=2, mlp_dim=32, dim_head=8)
DEC = dict(image_size=32, patch_size=8, num_classes=1, dim=12, depth=1, heads=2, mlp_dim=48, dim_head=4)
RATIO = 0.75


def _wrapper_params(spec_list, seed):
    rng = np.random.default_rng(seed)
    return {n: rng.standard_normal(s) * 0.3 for n, s in spec_list}


def _setup(seed=0, b=3):
    ecfg, dcfg = spec.make_config("vit", **ENC), spec.make_config("vit", **DEC)
    E, D = spec.init_params(ecfg, 1 + seed, True), spec.init_params(dcfg, 2 + seed, True)
    npat = 16
    Wm = _wrapper_params(RW.mae_param_spec(ecfg, npat + 1, DEC["dim"]), 3 + seed)
    Ws = _wrapper_params(RW.simmim_param_spec(ecfg), 4 + seed)
    rng = np.random.default_rng(5 + seed)
    img = rng.standard_normal((b, 32, 32, 3))
    perm = np.argsort(rng.uniform(size=(b, npat)), axis=-1)
    return ecfg, dcfg, E, D, Wm, Ws, img, perm


def test_num_masked_is_python_int_truncation():
    assert RW.num_masked(0.75, 64) == 48 and RW.num_masked(0.5, 49) == 24 and RW.num_masked(0.29, 100) == 28   # 28.999999999999996


def test_wrapper_param_specs():
    ecfg = spec.make_config("vit", **ENC)
    names = [n for n, _ in RW.mae_param_spec(ecfg, 17, 12)]
    assert names == ["enc_to_dec.kernel", "enc_to_dec.bias", "mask_token", "decoder_pos_emb.embeddings", "to_pixels.kernel", "to_pixels.bias"]
    same = dict(RW.mae_param_spec(ecfg, 17, ENC["dim"]))   # encoder_dim == decoder_dim: Identity, no Dense (mae.py:41)
    assert "enc_to_dec.kernel" not in same and same["decoder_pos_emb.embeddings"] == (17, 16) and same["to_pixels.kernel"] == (16, 192)
    assert RW.simmim_param_spec(ecfg) == [("mask_token", (16,)), ("to_pixels.kernel", (16, 192)), ("to_pixels.bias", (192,))]


def test_mae_oracle_against_explicit_index_arithmetic():
    ecfg, dcfg, E, D, Wm, _, img, perm = _setup()
    loss, pred, ge, gd, gw = RW.mae_forward_backward(ecfg, dcfg, E, D, Wm, img, perm, RATIO, literal_loss=True)
    nm = 12
    assert pred.shape == (3, nm, 192)
    assert abs(loss - np.mean(pred ** 2)) < 1e-12                       # mae.py:90 exactly as written
    patches = ref_numpy.patch_unfold(img, 8, 8)
    target = np.stack([patches[i][perm[i, :nm]] for i in range(3)])     # mae.py:65 by explicit loops
    loss2, pred2, *_ = RW.mae_forward_backward(ecfg, dcfg, E, D, Wm, img, perm, RATIO, literal_loss=False)
    assert np.allclose(pred, pred2) and abs(loss2 - np.mean((pred - target) ** 2)) < 1e-12
    # the last decoder_pos_emb row (the cls slot counted by mae.py:37) and unused model parts never receive gradient
    assert not gw["decoder_pos_emb.embeddings"][16].any() and gw["decoder_pos_emb.embeddings"][:16].any()
    assert not ge["cls_token"].any() and not ge["pos_embedding"][0, 0].any() and ge["pos_embedding"][0, 1:].any()
    assert not ge["mlp_head.kernel"].any() and not gd["patch_embedding.kernel"].any() and gd["transformer.0.mlp.fc1.kernel"].any()


def test_mae_loss_is_invariant_to_the_order_inside_each_index_set():
    ecfg, dcfg, E, D, Wm, _, img, perm = _setup(1)
    base = RW.mae_forward_backward(ecfg, dcfg, E, D, Wm, img, perm, RATIO, literal_loss=False)[0]
    p2 = perm.copy()
    p2[:, :12] = p2[:, :12][:, ::-1]      # reorder the masked set
    p2[:, 12:] = p2[:, 12:][:, [2, 0, 3, 1]]   # and the visible set: attention is permutation-equivariant, positions ride on the tokens
    assert abs(RW.mae_forward_backward(ecfg, dcfg, E, D, Wm, img, p2, RATIO, literal_loss=False)[0] - base) < 1e-12


def test_reference_tape_cut_only_removes_upstream_gradients():
    ecfg, dcfg, E, D, Wm, Ws, img, perm = _setup(2)
    full = RW.mae_forward_backward(ecfg, dcfg, E, D, Wm, img, perm, RATIO)
    cut = RW.mae_forward_backward(ecfg, dcfg, E, D, Wm, img, perm, RATIO, detach_like_reference=True)
    assert full[0] == cut[0]
    assert not cut[2]["patch_embedding.kernel"].any() and not cut[2]["pos_embedding"].any() and full[2]["patch_embedding.kernel"].any()
    for k in full[2]:
        if k.startswith("transformer."):
            assert np.array_equal(full[2][k], cut[2][k]), k
    midx = perm[:, :8]
    f = RW.simmim_forward_backward(ecfg, E, Ws, img, midx, 0.5)
    c = RW.simmim_forward_backward(ecfg, E, Ws, img, midx, 0.5, detach_like_reference=True)
    assert f[0] == c[0] and all(not g.any() for g in c[2].values()) and f[2]["transformer.0.attn.to_qkv.kernel"].any()
    assert np.array_equal(f[3]["to_pixels.kernel"], c[3]["to_pixels.kernel"]) and not c[3]["mask_token"].any() and f[3]["mask_token"].any()


def test_simmim_oracle_against_explicit_index_arithmetic():
    ecfg, _, E, _, _, Ws, img, perm = _setup(3)
    midx = perm[:, :8]
    loss, pred, ge, gw = RW.simmim_forward_backward(ecfg, E, Ws, img, midx, 0.5)
    patches = ref_numpy.patch_unfold(img, 8, 8)
    target = np.stack([patches[i][midx[i]] for i in range(3)])
    assert abs(loss - np.mean(np.abs(pred - target)) / 8) < 1e-12       # simmim.py:128
    assert abs(RW.simmim_forward_backward(ecfg, E, Ws, img, midx[:, ::-1], 0.5)[0] - loss) < 1e-12
    # tokens: masked slots are mask_token + pos, visible slots patch embeddings + pos (simmim.py:102-113)
    Et = ref_torch.to_torch(E)
    _, tok, pos = RW._patch_tokens(ecfg, Et, torch.tensor(img), ref_torch._ident)
    mixed = tok.numpy().copy()
    for i in range(3):
        for t in midx[i]:
            mixed[i, t] = Ws["mask_token"] + pos.numpy()[0, t]
    enc = ref_torch._transformer(torch.tensor(mixed), Et, ecfg, "transformer", 1, ref_torch._ident).numpy()
    ref_pred = np.stack([enc[i][midx[i]] for i in range(3)]) @ Ws["to_pixels.kernel"] + Ws["to_pixels.bias"]
    assert np.abs(ref_pred - pred).max() < 1e-10
    assert ge["pos_embedding"][0, 1:].any() and not ge["pos_embedding"][0, 0].any()


@pytest.mark.parametrize("which", ["mae", "simmim"])
def test_oracle_gradients_against_finite_differences(which):
    ecfg, dcfg, E, D, Wm, Ws, img, perm = _setup(4, b=2)
    if which == "mae":
        run = lambda W, Ee: RW.mae_forward_backward(ecfg, dcfg, Ee, D, W, img, perm, RATIO, literal_loss=False)
        W0, gi_w, gi_e = Wm, 4, 2
    else:
        run = lambda W, Ee: RW.simmim_forward_backward(ecfg, Ee, W, img, perm[:, :8], 0.5)
        W0, gi_w, gi_e = Ws, 3, 2
    out = run(W0, E)
    rng = np.random.default_rng(0)
    for group, name in (("w", "mask_token"), ("w", "to_pixels.kernel"), ("e", "patch_embedding.kernel"), ("e", "pos_embedding")):
        base = W0 if group == "w" else E
        g = out[gi_w][name] if group == "w" else out[gi_e][name]
        dirn = rng.standard_normal(base[name].shape)
        eps = 1e-5
        plus, minus = dict(base), dict(base)
        plus[name] = base[name] + eps * dirn
        minus[name] = base[name] - eps * dirn
        lp = run(plus, E)[0] if group == "w" else run(W0, plus)[0]
        lm = run(minus, E)[0] if group == "w" else run(W0, minus)[0]
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - float((g * dirn).sum())) <= 1e-6 * max(1.0, abs(fd)), name
