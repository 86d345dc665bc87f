"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Fixtures of the ORACLE (bisecting aids: per-block activations); the files that pin it to the reference are tests/golden/ref_*.npz (oracle/gen_ref_fixtures.py).

Writes the small golden fixtures under tests/golden/ from the ORACLE (the reference itself cannot be
imported here: TensorFlow is absent from the image and there is no network).  They freeze the oracle's
behaviour (numpy fp64 forward, torch-autograd fp64 gradients) so that (a) the two twins are pinned
against each other and against regressions, and (b) the GPU tests have seed-independent inputs.

    python -m oracle.gen_golden
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_numpy, ref_torch, spec  # noqa: E402
from util import CONFIGS, oracle_cfg  # noqa: E402

GOLDEN = ["vit_small", "vit_rect_mean", "vit_noproj", "deepvit_small", "cait_small"]


def make(name: str, b: int = 2):
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, seed=1, randomize_all=True)
    rng = np.random.Generator(np.random.PCG64(7))
    img = rng.standard_normal((b, *cfg["image_size"], 3)).astype(np.float32)
    dlogits = (rng.standard_normal((b, cfg["num_classes"])) / b).astype(np.float32)
    logits = ref_numpy.forward(cfg, P, img)
    logits_t, grads, dimg = ref_torch.forward_backward(cfg, P, img, dlogits, want_dimg=True)
    assert np.abs(logits - logits_t).max() < 1e-10, "oracle twins disagree"
    out = {"img": img, "dlogits": dlogits, "logits": logits, "dimg": dimg}
    # parameters are regenerated from (seed=1, randomize_all=True); a checksum pins the generator
    out["param_checksum"] = np.float64(sum(float(np.abs(v).sum()) for v in P.values()))
    for k, v in grads.items():
        out["grad/" + k] = v.astype(np.float32)
    return out


TOKENIZER_GEOMS = [(9, 7, 2, 3, 2), (8, 8, 1, 3, 1), (5, 5, 3, 7, 4), (6, 10, 2, 2, 3), (16, 16, 3, 7, 4)]   # (H, W, C, k, stride)
SHELL_KW = dict(image_size=(32, 48), patch_size=8, num_classes=7, dim=32)


def shell_middle(t):
    """The caller-supplied transformer of the efficient.ViT fixture: keeps the cls token and every other patch, doubled."""
    return t[:, ::2] * 2.0


def make_tokenizer_and_shell():
    """T2T tokenizer (t2t.py:42) and efficient.ViT shell (efficient.py:38-56) fixtures: inputs, outputs and VJPs from
    oracle/ref_efficient.py (loop form checked against the pad + im2col form while generating)."""
    import torch
    from oracle import ref_efficient as R
    rng = np.random.Generator(np.random.PCG64(11))
    out = {"geoms": np.asarray(TOKENIZER_GEOMS, dtype=np.int32)}
    for i, (H, W, C, k, s) in enumerate(TOKENIZER_GEOMS):
        x = rng.standard_normal((2, H, W, C)).astype(np.float32)
        y = R.extract_patches(x, k, s)
        xt = torch.tensor(x.astype(np.float64), requires_grad=True)
        yt = R.extract_patches_unfold(xt, k, s)
        assert np.array_equal(y, yt.detach().numpy().astype(np.float32)), "tokenizer formulations disagree"
        dy = rng.standard_normal(y.shape).astype(np.float32)
        (yt * torch.tensor(dy.astype(np.float64))).sum().backward()
        out.update({f"x{i}": x, f"y{i}": y, f"dy{i}": dy, f"dx{i}": xt.grad.numpy()})
    cfg = spec.make_config("vit", **SHELL_KW, depth=0, heads=1, mlp_dim=64, dim_head=64)
    P = spec.init_params(cfg, seed=11, randomize_all=True)
    img = rng.standard_normal((3, 32, 48, 3)).astype(np.float32)
    dl = (rng.standard_normal((3, 7)) / 2).astype(np.float32)
    logits, grads, dimg, tokens = R.shell_forward_backward(cfg, P, img, dl, shell_middle)
    out.update({"shell/img": img, "shell/dlogits": dl, "shell/logits": logits, "shell/dimg": dimg, "shell/tokens": tokens.astype(np.float32),
                "shell/param_checksum": np.float64(sum(float(np.abs(v).sum()) for v in P.values()))})
    for k_, v in grads.items():
        out["shell/grad/" + k_] = v.astype(np.float32)
    return out


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    d = make_tokenizer_and_shell()
    path = os.path.join(ROOT, "tests", "golden", "tokenizer_and_shell.npz")
    np.savez_compressed(path, **d)
    print("tokenizer_and_shell", os.path.getsize(path) // 1024, "KiB")
    for name in GOLDEN:
        d = make(name)
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **d)
        print(name, os.path.getsize(path) // 1024, "KiB", "logits std %.3f" % d["logits"].std())
