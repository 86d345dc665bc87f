#!/usr/bin/env python3
"""Per-shape table of the bf16 NT GEMM variants through the C ABI, one process (same box, same clocks):

    python tools/gemm_sweep.py [vitb|vitl|square] [iters]          # built-in shape lists
    VITX_GEMM_XP=1 python tools/gemm_sweep.py vitb 10 xp           # + timing experiments VITX_SWEEP_XP=0,1,2,4,6 (1: no DMA wait, 2: no DMA issue,
                                                                   #   4: no fragment reads in the K loop; results invalid)

epilogue codes of vitx_bench_gemm: 0 fp32 store, 1 bias + fp32 residual, 2 bias + GELU (two bf16 outputs), 3 bf16 store, 4 gelu' multiply + column sums.
Prints one line per (shape, epilogue, variant): ms, TFLOP/s, fraction of the 2516.6 TFLOP/s dense bf16 peak."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd"), os.path.join(ROOT, "tests")]
from util import make_engine_model  # noqa: E402
from vit_tensorflow import _native as N  # noqa: E402

PEAK = 2516.6
LISTS = {
    # (M, N, K, epilogues): the launches of one ViT-B/16 block at batch 256 (M = 256 * 197)
    "vitb": [(50432, 768, 768, (3, 1)), (50432, 2304, 768, (3,)), (50432, 3072, 768, (2, 4, 3)), (50432, 768, 3072, (1, 3)), (50432, 768, 2304, (3,))],
    "vitl": [(50432, 1024, 1024, (3, 1)), (50432, 3072, 1024, (3,)), (50432, 4096, 1024, (2, 4)), (50432, 1024, 4096, (1, 3))],
    # round 6, tail balancing: the full rounds of 256 x 256 tiles (main) and the rows of the last partial round (tail) as launches of their own
    "mainb": [(43520, 768, 768, (3, 1)), (43520, 768, 3072, (1, 3)), (43520, 768, 2304, (3,)), (49152, 3072, 768, (2, 4, 3)), (49920, 2304, 768, (3,))],
    "tailb": [(6912, 768, 768, (3, 1)), (6912, 768, 3072, (1, 3)), (6912, 768, 2304, (3,)), (1280, 3072, 768, (2, 4, 3)), (512, 2304, 768, (3,))],
    "deepvit": [(16640, 1024, 1024, (3, 1)), (16640, 3072, 1024, (3,)), (16640, 2048, 1024, (2, 4)), (16640, 1024, 2048, (1, 3))],
    "square": [(8192, 8192, 8192, (3,)), (4096, 4096, 4096, (3,))],
}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vitb"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    xp = len(sys.argv) > 3 and sys.argv[3] == "xp"
    # a variant code may carry a tail variant in bits 10..13 (tail balancing): 13 + (10 << 10) = 10253 is 256-row tiles + a 192 x 128 tail launch
    variants = [int(v) for v in os.environ.get("VITX_SWEEP_VARIANTS", "13,11,6,7").split(",")]
    m = make_engine_model("vit_bf16_small", "bf16", 1)
    m.build((1,))
    avg, err = C.c_float(), C.c_float()
    for (M, Nn, K, epis) in LISTS[which]:
        for epi in epis:
            for kern in variants:
                xs = tuple(int(v) for v in os.environ.get("VITX_SWEEP_XP", "0,1,2").split(","))
                for x in (xs if (xp and (kern & 15) >= 9) else (0,)):
                    N.check(N.lib().vitx_bench_gemm(m._handle, M, Nn, K, kern + 16 * x, epi, iters, C.byref(avg), C.byref(err)))
                    tf = 2.0 * M * Nn * K / (avg.value * 1e-3) / 1e12
                    print(f"M{M} N{Nn} K{K} epi {epi} variant {kern & 1023:2d} tail {kern >> 10:2d} xp {x}: {avg.value * 1e3:8.1f} us {tf:7.1f} TFLOP/s  frac {tf / PEAK:.3f}", flush=True)


if __name__ == "__main__":
    main()
