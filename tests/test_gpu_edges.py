"""Edge cases of the drop-in boundary on the GPU: batch geometry changes on one handle (the row-padding invariant has to be
re-established), batch 1 / empty batch, inputs the reference rejects, run-to-run determinism of every variant (no atomics in any
reduction), and the variant-independent results of the GEMM family through a full model."""
import numpy as np
import pytest

from oracle import ref_numpy, ref_torch, spec
from util import rel_max_err, gate, make_engine_model, oracle_cfg, rand_images
from vit_tensorflow import _native as N

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,compute,tol,gtol", [("vit_small", "fp32", 1e-4, 1e-3), ("vit_bf16_small", "bf16", 6e-3, 4e-2), ("cait_small", "fp32", 1e-4, 1e-3),
                                                   ("cait_bf16_small", "bf16", 2e-2, 9e-2), ("deepvit_bf16_small", "bf16", 2e-2, 9e-2)])
def test_batch_geometry_changes_on_one_handle(name, compute, tol, gtol):
    """b = 4 -> 1 -> 3 on the same handle, then a smaller image: each call must match the oracle -- logits AND every gradient (stale rows of the larger
    batch must not leak in; nothing that is not an activation may be touched by the re-zeroing of the row padding).
    Round 6: the bf16 CaiT rows are the regression test of the concatenated [to_q | to_kv] operand copies, which were registered as activation
    buffers and wiped by the first call with another geometry (found by tools/fuzz_configs.py "sequences"): until this round the backward of the
    changed geometry was only checked for finiteness."""
    from oracle import ref_torch
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, 1, randomize_all=True)
    m = make_engine_model(name, compute, 4, P)
    H, W = cfg["image_size"]
    ph, pw = cfg["patch_size"]
    for b, hw in ((4, None), (1, None), (3, None), (4, None), (2, (H - ph, W - pw))):
        img = rand_images(cfg, b, seed=10 + b, hw=hw)
        got = m(img, training=False)
        dl = (np.random.default_rng(b).standard_normal(got.shape) / b).astype(np.float32)
        ref, rg, _ = ref_torch.forward_backward(cfg, P, img, dl)
        assert got.shape == ref.shape
        gate(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()), tol, f"logits at b={b}", "logits")
        grads, _ = m.backward(dl)
        for k in rg:
            if np.asarray(rg[k]).size == 1 and compute == "bf16":
                continue   # (one-element head-mix tensors: a single, almost completely cancelling sum -- not resolvable in bf16, see tools/fuzz_configs.py)
            mix = k.endswith("reattn_weights") or "mix_heads" in k
            gate(rel_max_err(grads[k], rg[k]), (1.7e-1 if mix else gtol) if compute == "bf16" else gtol, f"{k} at b={b} image {img.shape[1:3]}", "gradients after a geometry change")


def test_batch_larger_than_max_batch():
    """The C ABI rejects a batch beyond the handle's plan (buffers are sized once); the Python mirror, like a Keras model, accepts
    any batch by rebuilding the plan with the weights kept."""
    import ctypes as C
    cfg = oracle_cfg("vit_small")
    P = spec.init_params(cfg, 1, randomize_all=True)
    m = make_engine_model("vit_small", "fp32", 2, P)
    m(rand_images(cfg, 2, seed=0), training=False)
    img3 = rand_images(cfg, 3, seed=1)
    out = np.empty((3, cfg["num_classes"]), np.float32)
    rc = N.lib().vitx_forward(m._handle, img3.ctypes.data_as(C.c_void_p), 3, img3.shape[1], img3.shape[2], 0, C.c_uint64(0),
                              out.ctypes.data_as(C.c_void_p))
    assert rc == N.ERR_INVALID
    got = m(img3, training=False)                     # mirror: grows the plan
    assert np.abs(got - ref_numpy.forward(cfg, P, img3)).max() <= 1e-4


def test_image_not_divisible_by_patch_is_rejected_with_the_reference_message():
    cfg = oracle_cfg("vit_small")
    m = make_engine_model("vit_small", "fp32", 2)
    bad = np.zeros((1, 60, 64, 3), np.float32)   # 60 % 16 != 0   (vit.py:136)
    with pytest.raises((N.VitxError, AssertionError)) as ei:
        m(bad, training=False)
    assert "divisible by the patch size" in str(ei.value)


def test_backward_without_forward_is_a_state_error():
    m = make_engine_model("vit_small", "fp32", 2)
    m.build((2,))
    with pytest.raises(N.VitxError) as ei:
        m.backward(np.zeros((2, 10), np.float32))
    assert ei.value.code == N.ERR_STATE


@pytest.mark.parametrize("name,compute", [("vit_bf16_small", "bf16"), ("deepvit_bf16_small", "bf16"), ("cait_bf16_small", "bf16"),
                                          ("deepvit_small", "fp32"), ("cait_small", "fp32")])
def test_forward_backward_is_bit_deterministic(name, compute):
    """Two identical steps give bit-identical logits and gradients: every reduction (split-K partials, LayerNorm / bias / mixing
    matrix sums) runs in a fixed order, nothing uses floating-point atomics."""
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, 1, randomize_all=True)
    m = make_engine_model(name, compute, 3, P)
    img = rand_images(cfg, 3, seed=5)
    dl = np.random.default_rng(1).standard_normal((3, cfg["num_classes"])).astype(np.float32)
    runs = []
    for _ in range(2):
        lg = m(img, training=False)
        grads, dimg = m.backward(dl, want_dimg=True)
        runs.append((lg, grads, dimg))
    assert np.array_equal(runs[0][0], runs[1][0])
    assert np.array_equal(runs[0][2], runs[1][2])
    for k in runs[0][1]:
        assert np.array_equal(runs[0][1][k], runs[1][1][k]), k


def test_gemm_variant_choice_does_not_change_results(monkeypatch):
    """Every NT variant accumulates each output element in the same K order: forcing different variants (and the measured
    choice) through a whole bf16 model gives bit-identical logits."""
    cfg = oracle_cfg("vit_bf16_small")
    P = spec.init_params(cfg, 1, randomize_all=True)
    img = rand_images(cfg, 3, seed=2)
    outs = []
    for kern in ("0", "2", "6", "14", "1"):
        monkeypatch.setenv("VITX_GEMM_KERNEL", kern)
        m = make_engine_model("vit_bf16_small", "bf16", 3, P)
        outs.append(m(img, training=False))
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)


@pytest.mark.parametrize("name,compute,tol", [("deepvit_82tok", "fp32", 2e-4), ("cait_82tok", "fp32", 2e-4), ("deepvit_82tok", "bf16", 2e-2),
                                              ("cait_82tok", "bf16", 2e-2)])   # bf16 (against the oracle with the same rounding points): observed logits 2.4e-3, gradients 9.1e-3
def test_head_axis_chains_beyond_64_keys(name, compute, tol):
    """Rows with more than 64 keys take the multi-sweep fused head-axis kernels (softmax / head mixing / LayerNorm over heads and
    their VJPs): logits and every gradient against the autograd twin."""
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, 1, randomize_all=True)
    m = make_engine_model(name, compute, 2, P)
    img = rand_images(cfg, 2, seed=7)
    dl = (np.random.default_rng(2).standard_normal((2, cfg["num_classes"])) / 2).astype(np.float32)
    logits = m(img, training=False)
    grads, _ = m.backward(dl)
    q = ref_torch.bf16_round if compute == "bf16" else None
    ref_logits, ref_grads, _ = ref_torch.forward_backward(cfg, P, img, dl, q=q)
    gate(np.abs(logits - ref_logits).max() / max(1.0, np.abs(ref_logits).max()), tol, "logits", "logits")
    for k, r in ref_grads.items():
        gate((np.abs(grads[k] - r).max() - 1e-6) / np.abs(r).max(), tol, k, "gradients")


def test_hip_graph_replay_equals_eager_steps():
    """vitx_graph_*: a captured training step (operand refresh, forward, CE gradient, backward, SGD) replayed as one launch leaves
    exactly the parameters that the same number of eager steps leaves (bit-identical: the same kernels in the same order)."""
    import ctypes as C
    import torch
    from vit_tensorflow import _native as N
    lib = N.lib()
    P = spec.init_params(oracle_cfg("vit_bf16_small"), 1, randomize_all=True)
    rng = np.random.default_rng(0)
    b = 3
    img = torch.tensor(rng.standard_normal((b, 64, 64, 3)).astype(np.float32), device="cuda:0")
    labels = torch.tensor(rng.integers(0, 10, b).astype(np.int32), device="cuda:0")
    outs = []
    for use_graph in (False, True):
        m = make_engine_model("vit_bf16_small", "bf16", b, P)
        h = m._ensure_handle(b)

        def step():
            N.check(lib.vitx_params_changed(h))
            N.check(lib.vitx_forward_dev(h, C.c_void_p(img.data_ptr()), b, 64, 64, 0, 0, None))
            N.check(lib.vitx_ce_loss_grad_dev(h, C.c_void_p(labels.data_ptr()), 1.0 / b, None))
            N.check(lib.vitx_backward_dev(h, None, None))
            N.check(lib.vitx_sgd_step(h, 1e-2, 0.0, 0.0))

        step()                                   # eager warm-up: first-use setup cannot be captured
        if use_graph:
            N.check(lib.vitx_graph_capture_begin(h))
            step()                               # recorded, not run
            g = C.c_void_p()
            N.check(lib.vitx_graph_capture_end(h, C.byref(g)))
            for _ in range(3):
                N.check(lib.vitx_graph_launch(h, g))
            N.check(lib.vitx_sync(h))
            N.check(lib.vitx_graph_destroy(g))
        else:
            for _ in range(3):
                step()
        m._device_newer = True
        outs.append(np.concatenate([w.reshape(-1) for w in m.get_weights()]))
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], np.concatenate([np.asarray(P[n], np.float32).reshape(-1) for n, _, _ in m._table]))


def test_optimizer_state_survives_a_handle_regrow(tmp_path):
    """The AdamW moments and step count live in the device handle; a call with batch > max_batch rebuilds the handle.  Training at
    b = 2, one evaluation at b = 4, then training again must continue exactly like an uninterrupted run."""
    cfg = oracle_cfg("vit_small")
    P = spec.init_params(cfg, 1, randomize_all=True)
    img = np.random.default_rng(0).standard_normal((4, 64, 64, 3)).astype(np.float32)
    dl = (np.random.default_rng(1).standard_normal((2, cfg["num_classes"])) / 2).astype(np.float32)

    def run(interrupt):
        m = make_engine_model("vit_small", "fp32", 2, P)
        for step in range(4):
            m(img[:2], training=False)
            m.backward(dl)
            m.apply_gradients("adamw", lr=1e-2, weight_decay=0.01)
            if interrupt and step == 1:
                m(img, training=False)          # b = 4 > max_batch = 2: the handle is rebuilt
                assert m._cfg.max_batch == 4
        return m.state_dict()

    a, b = run(False), run(True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    # save_weights / load_weights agree on the file name with or without the .npz suffix (np.savez appends it)
    m = make_engine_model("vit_small", "fp32", 2, P)
    for name in ("ckpt", "ckpt2.npz"):
        m.save_weights(str(tmp_path / name))
        m2 = make_engine_model("vit_small", "fp32", 2)
        m2.load_weights(str(tmp_path / name))
        assert all(np.array_equal(v, m2.state_dict()[k]) for k, v in m.state_dict().items())
    # the Keras get_weights()-order list form (what np.savez(path, *keras_model.get_weights()) writes on the TensorFlow side)
    m.save_weights(str(tmp_path / "klist"), format="keras_list")
    with np.load(str(tmp_path / "klist.npz")) as z:
        assert z.files == [f"arr_{i}" for i in range(len(m.get_weights()))] and z["arr_0"].shape == m.weights[0].shape
    m3 = make_engine_model("vit_small", "fp32", 2)
    m3.load_weights(str(tmp_path / "klist"))
    assert all(np.array_equal(v, m3.state_dict()[k]) for k, v in m.state_dict().items())


@pytest.mark.parametrize("name", ["vit_small", "deepvit_small", "cait_small", "cfg2_vit_b16"])
def test_fp32_mfma_gemm_is_bit_identical_to_the_scalar_fma_gemm(name, monkeypatch):
    """FP32_PARITY mode runs its GEMMs on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, gemm_f32_mfma.hip).  That instruction is a
    k-ordered fmaf chain, i.e. the arithmetic of the scalar kernel it replaces: logits and every gradient must come out
    bit-identical with VITX_F32_MFMA=0 (scalar FMA kernel) and =1 (matrix pipe) -- Dense layers, their VJPs (all three stride
    patterns) and the batched attention products."""
    from util import CONFIGS
    kw = dict(CONFIGS[name][1])
    if name == "cfg2_vit_b16":
        kw["depth"] = 1
    v = CONFIGS[name][0]
    cfg = spec.make_config(v, **kw)
    P = spec.init_params(cfg, 1, randomize_all=True)
    img = rand_images(cfg, 2)
    dl = (np.random.default_rng(5).standard_normal((2, cfg["num_classes"])) / 2).astype(np.float32)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("VITX_F32_MFMA", flag)
        if v == "vit":
            from vit_tensorflow import ViT as cls
        elif v == "deepvit":
            from vit_tensorflow.deepvit import DeepViT as cls
        else:
            from vit_tensorflow.cait import CaiT as cls
        m = cls(**kw, compute="fp32", max_batch=2, seed=0)
        m.load_state_dict({k: a.astype(np.float32) for k, a in P.items()})
        lg = m(img, training=False)
        g, dimg = m.backward(dl, want_dimg=True)
        outs.append((lg, g, dimg))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][2], outs[1][2])
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k
