#!/usr/bin/env python3
"""vitx_check_gemm over a list of shapes for one NT variant:  python tools/check_variant.py <variant> <epilogue> M,N,K [M,N,K ...]"""
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
v, epi = int(sys.argv[1]), int(sys.argv[2])
for s in sys.argv[3:]:
    M, Nn, K = (int(x) for x in s.split(","))
    N.check(N.lib().vitx_check_gemm(m._handle, 0, M, Nn, K, v, epi, errs))
    print(f"variant {v} epi {epi} M{M} N{Nn} K{K}: err {errs[0]:.3e} {errs[1]:.3e}", flush=True)
