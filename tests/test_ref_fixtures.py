"""CPU tier: the oracle against fixtures produced by the REFERENCE'S OWN SOURCE (tests/golden/ref_*.npz, written by
oracle/gen_ref_fixtures.py, which imports /root/reference/vit_tensorflow/{vit,deepvit,cait,parallel_vit,vit_with_patch_merger}.py
unmodified under oracle/tf_shim).  This is what pins parity: oracle/ref_numpy.py and oracle/ref_torch.py must reproduce the
reference's logits, every variable's gradient and d(img) to float64 rounding.  Where /root/reference is present (this container,
not the GPU box) the reference is also re-executed live and compared with the committed files, so the fixtures cannot drift from
the source they claim to come from."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gen_ref_fixtures as G
from oracle import ref_numpy, ref_torch, spec

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F64_TOL = 1e-12


def _load(case):
    return np.load(os.path.join(GOLDEN_DIR, f"ref_{case}.npz"))


def _params(case, z):
    cfg = G.oracle_cfg_of(case)
    P = spec.init_params(cfg, seed=int(z["param_seed"]), randomize_all=True)
    assert abs(sum(float(np.abs(v).sum()) for v in P.values()) - float(z["param_checksum"])) < 1e-6
    return cfg, P


@pytest.mark.parametrize("case", list(G.CASES))
def test_oracle_reproduces_reference_fixture(case):
    z = _load(case)
    cfg, P = _params(case, z)
    assert z["logits"].shape == (2, cfg["num_classes"])
    if cfg["variant"] != "patch_merger" and cfg["num_parallel_branches"] == 1:   # ref_numpy covers vit / deepvit / cait
        assert np.abs(ref_numpy.forward(cfg, P, z["img"]) - z["logits"]).max() <= F64_TOL
    logits, grads, dimg = ref_torch.forward_backward(cfg, P, z["img"], z["dlogits"], want_dimg=True)
    assert np.abs(logits - z["logits"]).max() <= F64_TOL
    names = [n for n, _, _ in spec.param_spec(cfg)]
    assert sorted("grad/" + n for n in names) == sorted(k for k in z.files if k.startswith("grad/"))
    for n in names:
        ref = z["grad/" + n]
        assert np.abs(ref).max() > 0, n          # every variable of the reference received a gradient
        assert np.abs(grads[n] - ref).max() <= F64_TOL * max(1.0, np.abs(ref).max()), n
    assert np.abs(dimg - z["dimg"]).max() <= F64_TOL * max(1.0, np.abs(z["dimg"]).max())


@pytest.mark.parametrize("case", list(G.WIDE_CASES))
def test_oracle_reproduces_reference_fixture_at_baseline_widths(case):
    """BASELINE.json widths (ViT-B/16 224, DeepViT cfg4, CaiT cfg5) at depth 2, batch 2: logits in full, gradients by digest."""
    z = _load(case)
    cfg, P = _params(case, z)
    logits, grads, dimg = ref_torch.forward_backward(cfg, P, z["img"], z["dlogits"], want_dimg=True)
    assert np.abs(logits - z["logits"]).max() <= 1e-11
    for n, _, _ in spec.param_spec(cfg):
        f = grads[n].reshape(-1)
        scale = max(1.0, float(z["gabs/" + n]))
        assert abs(f.sum() - float(z["gsum/" + n])) <= 1e-10 * scale, n
        assert abs(np.abs(f).sum() - float(z["gabs/" + n])) <= 1e-10 * scale, n
        assert np.abs(f[::max(1, f.size // 256)][:256] - z["gsample/" + n]).max() <= 1e-6 * max(1.0, np.abs(f).max()), n
    assert np.abs(dimg.reshape(-1)[::997] - z["dimg_sample"]).max() <= 1e-6 * max(1.0, np.abs(dimg).max())


@pytest.mark.skipif(not os.path.isdir(G.REF), reason="/root/reference is not present on this machine (GPU box)")
@pytest.mark.parametrize("case", ["vit_small", "deepvit_small", "cait_small", "parallel_vit_2", "patch_merger_default"])
def test_committed_fixture_is_what_the_reference_source_produces(case):
    """Re-run the reference's source under the shim in a fresh interpreter (the shim registers a fake `tensorflow` in sys.modules,
    which must not leak into this process) and compare with the committed file bit for bit."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from oracle import gen_ref_fixtures as G\n"
        "d = G.make(%r); z = np.load(%r)\n"
        "assert sorted(d) == sorted(z.files), (sorted(d), sorted(z.files))\n"
        "bad = [k for k in d if not np.array_equal(np.asarray(d[k]), z[k])]\n"
        "assert not bad, bad\n"
        "import vit, inspect; assert inspect.getsourcefile(vit).startswith(%r)\n"
        "print('OK')\n" % (ROOT, os.path.join(ROOT, "tests"), case, os.path.join(GOLDEN_DIR, f"ref_{case}.npz"), os.path.dirname(G.REF))
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir(G.REF), reason="/root/reference is not present on this machine (GPU box)")
def test_committed_mpp_fixture_is_what_the_reference_source_produces():
    """tests/golden/ref_mpp_vit.npz = the reference's own mpp.py (MPP.call mpp.py:166-218, MPPLoss :90-131) re-run under the shim in a fresh
    interpreter, bit for bit."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from oracle import gen_ref_fixtures as G\n"
        "d = G.make_mpp('mpp_vit'); z = np.load(%r)\n"
        "assert sorted(d) == sorted(z.files), (sorted(d), sorted(z.files))\n"
        "bad = [k for k in d if not np.array_equal(np.asarray(d[k]), z[k])]\n"
        "assert not bad, bad\n"
        "import mpp, inspect; assert inspect.getsourcefile(mpp).startswith(%r)\n"
        "print('OK')\n" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(GOLDEN_DIR, "ref_mpp_vit.npz"), os.path.dirname(G.REF))
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_mpp_oracle_reproduces_reference_fixture():
    """oracle/ref_wrappers.py:mpp_forward (literal form) against what the reference's own mpp.py produced under the shim on the mask it drew:
    the loss AS WRITTEN (mpp.py:125 passes (predictions, labels) to softmax_cross_entropy_with_logits(labels, logits); :185,190 write the
    replacements into `.numpy()` copies), every variable's gradient, and which variables have none (mask_token, the encoder's mlp_head)."""
    import torch
    from oracle import ref_wrappers as RW
    z = _load("mpp_vit")
    ekw, wkw = G.MPP_CASES["mpp_vit"]
    ecfg = spec.make_config("vit", **ekw)
    E = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in spec.init_params(ecfg, int(z["enc_seed"]), randomize_all=True).items()}
    rng = np.random.Generator(np.random.PCG64(int(z["wrap_seed"])))
    Wp = {n: torch.tensor(0.3 * rng.standard_normal(shape), dtype=torch.float64, requires_grad=True) for n, shape in RW.mpp_param_spec(ecfg, wkw["output_channel_bits"])}
    img = torch.tensor(np.asarray(z["img"], np.float64))
    assert z["indices"].shape[1] == RW.mpp_num_masked(wkw["mask_prob"], (ekw["image_size"] // ekw["patch_size"]) ** 2)
    loss, _ = RW.mpp_forward(ecfg, E, Wp, img, z["indices"], wkw["output_channel_bits"], literal=True)
    assert abs(float(loss) - float(z["loss"])) <= F64_TOL * abs(float(z["loss"]))
    loss.backward()
    got = {**{"encoder." + k: v for k, v in E.items()}, **Wp}
    names = [k[5:] for k in z.files if k.startswith("grad/")]
    assert sorted(names) == sorted(got), sorted(set(names) ^ set(got))
    with_grad = 0
    for n in names:
        g = got[n].grad
        if not bool(z["has_grad/" + n]):
            assert g is None or not g.numpy().any(), n
            continue
        with_grad += 1
        ref = z["grad/" + n]
        assert np.abs(g.numpy().reshape(ref.shape) - ref).max() <= F64_TOL * max(1.0, np.abs(ref).max()), n
    assert with_grad == 28
    # the intended form (literal=False) is a different loss: a proper cross-entropy, finite and positive
    l2, _ = RW.mpp_forward(ecfg, {k: v.detach() for k, v in E.items()}, {k: v.detach() for k, v in Wp.items()}, img, z["indices"], wkw["output_channel_bits"], literal=False)
    assert np.isfinite(float(l2)) and float(l2) > 0


def test_shim_does_not_leak_into_this_process():
    assert "tensorflow" not in sys.modules or not getattr(sys.modules["tensorflow"], "__vitx_shim__", False)


@pytest.mark.parametrize("case", list(G.T2T_CASES))
def test_t2t_oracle_reproduces_reference_fixture(case):
    """oracle/ref_t2t.py (T2TViT, t2t.py:49-122) against the fixture the reference's own t2t.py produced under the shim."""
    from oracle import ref_t2t
    z = _load(case)
    cfg = ref_t2t.make_config(**G.T2T_CASES[case])
    P = ref_t2t.init_params(cfg, seed=int(z["param_seed"]))
    assert abs(sum(float(np.abs(v).sum()) for v in P.values()) - float(z["param_checksum"])) < 1e-6
    logits, grads, dimg = ref_t2t.forward_backward(cfg, P, z["img"], z["dlogits"])
    assert np.abs(logits - z["logits"]).max() <= F64_TOL
    for n, _, _ in ref_t2t.param_spec(cfg):
        ref = z["grad/" + n]
        assert np.abs(ref).max() > 0, n
        assert np.abs(grads[n] - ref).max() <= F64_TOL * max(1.0, np.abs(ref).max()), n
    assert np.abs(dimg - z["dimg"]).max() <= F64_TOL * max(1.0, np.abs(z["dimg"]).max())


def _distill_oracle(case):
    """(cfg, P, student_fn) of a distill fixture's student, in the oracle's terms."""
    from oracle import ref_distill, ref_t2t
    kind, kw, _ = G.DISTILL_CASES[case]
    if kind == "vit":
        cfg = spec.make_config("vit", **kw)
        return cfg, spec.init_params(cfg, seed=1, randomize_all=True), (lambda c, P, im, tok: ref_distill.student_forward(c, P, im, tok))
    cfg = ref_t2t.make_config(**kw)
    return cfg, ref_t2t.init_params(cfg, seed=1), ref_t2t.student_forward


@pytest.mark.parametrize("case", list(G.DISTILL_CASES))
def test_distill_oracle_reproduces_reference_fixture(case):
    """oracle/ref_distill.py (+ ref_t2t.student_forward) against what the reference's own distill.py produced under the shim:
    Distillable*.call(img, distill_token) with its VJP, and DistillWrapper's per-image loss (soft mode, as written: Keras'
    KLDivergence clips the log-probabilities it is handed) with d(sum loss)/d(every variable)."""
    import torch
    from oracle import ref_distill
    z = _load(case)
    cfg, P, student_fn = _distill_oracle(case)
    assert abs(sum(float(np.abs(v).sum()) for v in P.values()) - float(z["param_checksum"])) < 1e-6
    Pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
    img = torch.tensor(z["img"].astype(np.float64))
    tok = torch.tensor(z["call/token"], requires_grad=True)
    logits, dt = student_fn(cfg, Pt, img, tok)
    assert np.abs(logits.detach().numpy() - z["call/logits"]).max() <= F64_TOL
    assert np.abs(dt.detach().numpy() - z["call/distill_tokens"]).max() <= F64_TOL
    ((logits * torch.tensor(z["call/dlogits"].astype(np.float64))).sum() + (dt * torch.tensor(z["call/d_distill_tokens"].astype(np.float64))).sum()).backward()
    assert np.abs(tok.grad.numpy() - z["call/grad_token"]).max() <= F64_TOL * max(1.0, np.abs(z["call/grad_token"]).max())
    for n, v in Pt.items():
        ref = z["call/grad/" + n]
        assert np.abs(v.grad.numpy() - ref).max() <= F64_TOL * max(1.0, np.abs(ref).max()), n
    # the wrapper
    Pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
    Wd = {n: torch.tensor(z["wrap/param/" + n], dtype=torch.float64, requires_grad=True) for n, _ in ref_distill.wrapper_param_spec(cfg["dim"], cfg["num_classes"])}
    loss, _, _ = ref_distill.wrapper_loss(cfg, Pt, Wd, img, torch.tensor(z["wrap/labels"].astype(np.float64)),
                                          torch.tensor(z["wrap/teacher_logits"].astype(np.float64)), temperature=float(z["wrap/temperature"]),
                                          alpha=float(z["wrap/alpha"]), hard=False, literal_loss=True, student_fn=student_fn)
    assert np.abs(loss.detach().numpy() - z["wrap/loss"]).max() <= 1e-10 * np.abs(z["wrap/loss"]).max()
    loss.sum().backward()
    for n, v in Wd.items():
        ref = z["wrap/grad/" + n]
        g = v.grad.numpy() if v.grad is not None else np.zeros_like(ref)
        assert np.abs(g - ref).max() <= F64_TOL * max(1.0, np.abs(ref).max()), n
    for n, v in Pt.items():
        ref = z["wrap/grad/student." + n]
        assert np.abs(v.grad.numpy() - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), n


@pytest.mark.parametrize("case", list(G.MIM_CASES))
def test_mim_oracle_reproduces_reference_fixture(case):
    """oracle/ref_wrappers.py (MAE.call mae.py:47-92, SimMIM.call simmim.py:86-130) against what the reference's own mae.py /
    simmim.py produced under the shim on the indices the reference itself drew: the loss as written (mae.py:90 squares the
    prediction alone), every variable's gradient, and the tape cut of the `.numpy()` indexing (mae.py:62, simmim.py:119) --
    variables upstream of it have NO gradient in the reference, which `detach_like_reference=True` reproduces."""
    from oracle import ref_wrappers as RW
    z = _load(case)
    kind, ekw, wkw = G.MIM_CASES[case]
    ecfg = spec.make_config("vit", **ekw)
    E = spec.init_params(ecfg, int(z["enc_seed"]), randomize_all=True)
    Wp = G.mim_wrapper_params(kind, ecfg, wkw, int(z["wrap_seed"]))
    if kind == "mae":
        dcfg = G.mim_decoder_cfg(ekw, wkw)
        D = spec.init_params(dcfg, int(z["dec_seed"]), randomize_all=True)
        loss, _, ge, gd, gw = RW.mae_forward_backward(ecfg, dcfg, E, D, Wp, z["img"], z["indices"], float(z["masking_ratio"]),
                                                      literal_loss=True, detach_like_reference=True)
        got = {**{"encoder." + k: v for k, v in ge.items()}, **{"decoder." + k: v for k, v in gd.items() if k.startswith("transformer.")}, **gw}
    else:
        loss, _, ge, gw = RW.simmim_forward_backward(ecfg, E, Wp, z["img"], z["indices"], float(z["masking_ratio"]), detach_like_reference=True)
        got = {**{"encoder." + k: v for k, v in ge.items()}, **gw}
    assert abs(loss - float(z["loss"])) <= F64_TOL * abs(float(z["loss"]))
    names = [k[5:] for k in z.files if k.startswith("grad/")]
    assert sorted(names) == sorted(got), sorted(set(names) ^ set(got))
    with_grad = 0
    for n in names:
        ref = z["grad/" + n]
        if not bool(z["has_grad/" + n]):
            assert not np.asarray(got[n]).any(), n          # cut off from the tape in the reference: the oracle must agree
            continue
        with_grad += 1
        assert np.abs(got[n] - ref).max() <= F64_TOL * max(1.0, np.abs(ref).max()), n
    assert with_grad == {"mae_vit": 39, "mae_same_dim": 48, "simmim_vit": 2}[case]


@pytest.mark.skipif(not os.path.isdir(G.REF), reason="/root/reference is not present on this machine (GPU box)")
def test_t2t_weight_order_follows_keras_own_variables_then_sublayers():
    """Keras lists a Model's weights as its OWN tf.Variables first, then its sublayers' in attribute-creation order (Layer.weights =
    own trainable weights + children's).  Derived from the reference's source, not from the oracle: the order in which T2TViT.__init__
    (t2t.py:49-95) assigns its attributes, with tf.Variable attributes moved to the front."""
    import ast
    from oracle import ref_t2t
    src = open(os.path.join(G.REF, "t2t.py")).read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "T2TViT")
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    attrs = []
    for st in ast.walk(init):
        if isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Attribute) and getattr(st.targets[0].value, "id", "") == "self":
            is_var = isinstance(st.value, ast.Call) and ast.unparse(st.value.func).endswith("Variable")
            attrs.append((st.lineno, st.targets[0].attr, is_var))
    attrs.sort()
    own = [a for _, a, v in attrs if v]
    layers = list(dict.fromkeys(a for _, a, v in attrs if not v))   # (self.transformer is assigned in both arms of an if)
    assert own == ["pos_embedding", "cls_token"], own
    cfg = ref_t2t.make_config(image_size=32, num_classes=7, dim=32, depth=2, heads=2, mlp_dim=64, dim_head=16, t2t_layers=((3, 2), (3, 2), (3, 2)))
    groups = []
    for n, _, _ in ref_t2t.param_spec(cfg):
        g = n.split(".")[0]
        if not groups or groups[-1] != g:
            groups.append(g)
    expect = own + [a for a in layers if a in ("patch_embedding", "transformer", "mlp_head")]
    assert groups == expect, (groups, expect)
    # and the random draw behind the committed fixtures did not move with the listing order
    P = ref_t2t.init_params(cfg, seed=1)
    z = np.load(os.path.join(GOLDEN_DIR, "ref_t2t_small.npz"))
    assert abs(sum(float(np.abs(v).sum()) for v in P.values()) - float(z["param_checksum"])) < 1e-6
