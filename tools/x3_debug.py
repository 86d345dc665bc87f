#!/usr/bin/env python3
"""bf16x3 vs fp32 engine on one fixture: per-tensor gradient error (debug aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd"), os.path.join(ROOT, "tests")]
import test_gpu_ref_fixtures as T
from oracle import spec
case = sys.argv[1] if len(sys.argv) > 1 else "vit_small"
z, cfg, P = T._case(case)
out = {}
for comp in ("fp32", "bf16x3"):
    m = T._model(case, comp, 2, P)
    logits = m(z["img"], training=True)
    grads, dimg = m.backward(z["dlogits"], want_dimg=True)
    out[comp] = (logits, grads, dimg)
print("logits", np.abs(out["fp32"][0] - out["bf16x3"][0]).max())
for n, shp, _ in spec.param_spec(cfg):
    a, b = np.asarray(out["fp32"][1][n]), np.asarray(out["bf16x3"][1][n])
    e = np.abs(a - b).max() / (np.abs(a).max() + 1e-30)
    flag = "  <-- BAD" if e > 1e-3 else ""
    print(f"{n:50s} {str(a.shape):16s} rel {e:.3e}{flag}")
    if flag and a.ndim == 2:
        d = np.abs(a - b)
        rows = np.where(d.max(1) > 1e-3 * np.abs(a).max())[0]; cols = np.where(d.max(0) > 1e-3 * np.abs(a).max())[0]
        print("     bad rows", rows[:8], "...", rows[-3:], len(rows), " bad cols", cols[:8], "...", cols[-3:], len(cols))
