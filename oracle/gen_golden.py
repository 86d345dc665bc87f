"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  PARITY UNPINNED (see oracle/spec.py).

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


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for name in GOLDEN:
        d = make(name)
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **d)
        print(name, os.path.getsize(path) // 1024, "KiB", "logits std %.3f" % d["logits"].std())
