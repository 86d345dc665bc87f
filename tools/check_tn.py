#!/usr/bin/env python3
"""vitx_check_gemm, kind 1 (weight-gradient kernel with the engine's split rule + reduction vs the fp32-FMA kernel):  python tools/check_tn.py M,N,K ..."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd"), os.path.join(ROOT, "tests")]
from util import make_engine_model  # noqa: E402
from vit_tensorflow import _native as N  # noqa: E402

m = make_engine_model("vit_bf16_small", "bf16", 1)
m.build((1,))
errs = (C.c_float * 2)()
for s in sys.argv[1:]:
    M, Nn, K = (int(x) for x in s.split(","))
    N.check(N.lib().vitx_check_gemm(m._handle, 1, M, Nn, K, 0, 0, errs))
    print(f"tn M{M} N{Nn} K{K}: err {errs[0]:.3e}", flush=True)
