#!/usr/bin/env python3
"""Stand-alone GEMM micro-benchmark through the C ABI (for rocprofv3 --pmc runs):
    python tools/gemm_bench.py M N K kernel epilogue iters"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd"), os.path.join(ROOT, "tests")]
from util import make_engine_model  # noqa: E402
from vit_tensorflow import _native as N  # noqa: E402

M, Nn, K, kern, epi, iters = (int(x) for x in sys.argv[1:7])
m = make_engine_model("vit_bf16_small", "bf16", 1)
m.build((1,))
avg, err = C.c_float(), C.c_float()
N.check(N.lib().vitx_bench_gemm(m._handle, M, Nn, K, kern, epi, iters, C.byref(avg), C.byref(err)))
print(f"M{M} N{Nn} K{K} kernel {kern} epi {epi}: {avg.value:.4f} ms {2.0 * M * Nn * K / (avg.value * 1e-3) / 1e12:.1f} TFLOP/s")
