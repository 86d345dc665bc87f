#!/usr/bin/env python3
"""Correctness + timing of the wave-group ping-pong GEMM (variant 12) through the C ABI: device-side comparison with the fp32-FMA kernel
(vitx_check_gemm) on small / odd tile counts and the ViT-B launch shapes, then per-shape times next to the pipelined variants."""
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
avg, err = C.c_float(), C.c_float()
epis = [int(x) for x in os.environ.get("PP_EPIS", "3,2").split(",")]
bad = 0
if "--no-check" not in sys.argv:
    shapes = [(256, 128, 768), (256, 256, 768), (512, 256, 768), (768, 128, 768), (768, 384, 1024), (2560, 768, 768), (256 * 33, 128 * 5, 640),
              (50432, 768, 768), (50432, 3072, 768), (50432, 768, 3072), (50432, 2304, 768)]
    for (M, Nn, K) in shapes:
        for epi in epis:
            N.check(N.lib().vitx_check_gemm(m._handle, 0, M, Nn, K, int(os.environ.get("PP_CHECK_VARIANT", "12")), epi, errs))
            ok = errs[0] <= 1.1e-2 and (epi != 2 or errs[1] <= 1.5e-2)
            bad += not ok
            print(f"check M{M} N{Nn} K{K} epi {epi}: err {errs[0]:.3e} {errs[1]:.3e} {'ok' if ok else 'FAIL'}", flush=True)
if "--no-time" not in sys.argv:
    for (M, Nn, K, es) in [(50432, 3072, 768, (2, 3)), (50432, 2304, 768, (3,)), (50432, 768, 768, (3,)), (50432, 768, 3072, (3,)), (50432, 768, 2304, (3,)),
                           (8192, 8192, 8192, (3,))]:
        for epi in es:
            if epi not in epis:
                continue
            for kern in [int(v) for v in os.environ.get("PP_VARIANTS", "12,13,14,15").split(",")]:
                N.check(N.lib().vitx_bench_gemm(m._handle, M, Nn, K, kern, epi, 10, C.byref(avg), C.byref(err)))
                tf = 2.0 * M * Nn * K / (avg.value * 1e-3) / 1e12
                print(f"time M{M} N{Nn} K{K} epi {epi} variant {kern:2d}: {avg.value * 1e3:8.1f} us {tf:7.1f} TFLOP/s", flush=True)
sys.exit(1 if bad else 0)
