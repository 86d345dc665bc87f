#!/usr/bin/env python3
"""GPU-box diagnostics: each stage is run in its own subprocess by tools/gpu_round.sh so that a device fault in
one stage cannot hide the results of the others.  Writes human-readable logs (stdout) that are merged back
through gpurun_out/.

    python tools/gpu_diag.py env | gemm | parity_fp32 | parity_bf16 | grads | bench_quick
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vit-tensorflow_amd"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402


def stage_env():
    import torch
    print("torch", torch.__version__, "cuda available", torch.cuda.is_available())
    if torch.cuda.is_available():
        p = torch.cuda.get_device_properties(0)
        print("device:", p.name, "CUs", p.multi_processor_count, "mem GB", round(p.total_memory / 2 ** 30, 1))
    try:
        out = subprocess.run("rocminfo | grep -E 'Name:|Compute Unit|Max Clock|Wavefront' | head -40", shell=True, capture_output=True, text=True).stdout
        print(out)
    except Exception as ex:
        print("rocminfo failed", ex)
    print(subprocess.run("nproc; lscpu | grep -E 'Model name|Socket|Core' | head -5", shell=True, capture_output=True, text=True).stdout)


def _model(name, compute, b, P=None):
    from util import make_engine_model
    return make_engine_model(name, compute=compute, max_batch=b, params=P)


def stage_gemm():
    from vit_tensorflow import _native as N
    m = _model("vit_bf16_small", "bf16", 1)
    m.build((1,))
    lib = N.lib()
    avg, err = C.c_float(), C.c_float()
    print("== correctness (vs fp32-FMA generic kernel on the same bf16 operands)")
    for kern in (1, 2, 3, 5, 6, 7, 9, 10, 13, 14, 15, 0):
        for (M, Nn, K) in [(256, 256, 64), (256, 256, 128), (256, 256, 192), (300, 200, 192), (3000, 2304, 64), (50432, 768, 128), (1000, 768, 768), (2048, 1024, 4096)]:
            rc = lib.vitx_bench_gemm(m._handle, M, Nn, K, kern, 0, 1, C.byref(avg), C.byref(err))
            print(f"kernel {kern} M{M} N{Nn} K{K}: rc {rc} max_abs_err {err.value:.4g}")
    print("== throughput (random bf16 operands), epilogue 0=f32 store 1=bias+resid f32 2=bias+gelu 2xbf16 3=bf16 store")
    print("   kernel code: 1=128^2 2=256^2 3=256x128 5=320x256 | persistent: 6/7 lockstep 256^2/320x256, 9/10 software-pipelined, 13/14/15 + DMA pieces spread | 0 = measured choice; +256 direct epilogue, +512 LDS-staged")
    shapes = [(50432, 768, 768), (50432, 2304, 768), (50432, 3072, 768), (50432, 768, 3072), (8192, 8192, 8192), (4096, 4096, 4096)]
    for (M, Nn, K) in shapes:
        for kern in (2, 6, 13, 14, 5, 7, 15, 0):
            for epi in (0, 1, 2, 3):
                big = (M, Nn, K) in [(8192, 8192, 8192), (4096, 4096, 4096)]
                if big and (epi not in (0, 3) or kern >= 16 and kern < 256):
                    continue
                if (kern & 255) >= 16 and epi in (0, 3):
                    continue
                rc = lib.vitx_bench_gemm(m._handle, M, Nn, K, kern, epi, 20, C.byref(avg), C.byref(err))
                tf = 2.0 * M * Nn * K / (avg.value * 1e-3) / 1e12 if rc == 0 else float("nan")
                print(f"M{M} N{Nn} K{K} kernel {kern} epi {epi}: {avg.value:.4f} ms  {tf:.1f} TFLOP/s  rc {rc}", flush=True)


def stage_gemmx():
    """timing experiments on the software-pipelined kernel (results wrong, time only): +16 = no DMA wait, +32 = no DMA issue in
    the K loop; run once with random and once with zero-filled operands (clock / power effect)"""
    from vit_tensorflow import _native as N
    m = _model("vit_bf16_small", "bf16", 1)
    m.build((1,))
    lib = N.lib()
    avg, err = C.c_float(), C.c_float()
    for zero in (0, 1):
        if zero:
            os.environ["VITX_BENCH_ZERO"] = "1"
        for (M, Nn, K) in [(8192, 8192, 8192), (50432, 768, 3072), (50432, 3072, 768)]:
            for kern in (2 + 512, 6, 9, 9 + 16, 9 + 32, 14, 14 + 32):
                rc = lib.vitx_bench_gemm(m._handle, M, Nn, K, kern, 3, 20, C.byref(avg), C.byref(err))
                tf = 2.0 * M * Nn * K / (avg.value * 1e-3) / 1e12 if rc == 0 else float("nan")
                print(f"zero={zero} M{M} N{Nn} K{K} kernel {kern}: {avg.value:.4f} ms  {tf:.1f} TFLOP/s  rc {rc}", flush=True)


def _bisect(m, cfg, P, img, variant_prefixes):
    from oracle import ref_numpy
    acts = {}
    ref_numpy.forward(cfg, P, img, acts)
    b = img.shape[0]

    def cmp(tag, got, ref):
        ref = np.asarray(ref, np.float64).reshape(-1)
        got = got.reshape(-1)[: ref.size]
        print(f"   {tag:28s} max|d| {np.abs(got - ref).max():.3e}  ref absmax {np.abs(ref).max():.3e}")

    try:
        cmp("patches", m.debug_read("patches", 0), acts["patches"])
        cmp("embed (x_in[0])", m.debug_read("x_in", 0), acts["embed"])
        layer = 0
        for prefix, depth in variant_prefixes:
            for i in range(depth):
                cmp(f"{prefix}.{i}.attn_out", m.debug_read("attn_out", layer), acts[f"{prefix}.{i}.attn.attn_out"])
                cmp(f"{prefix}.{i}.out", m.debug_read("x_out", layer), acts[f"{prefix}.{i}.out"])
                layer += 1
    except Exception:
        traceback.print_exc()


def _prefixes(cfg):
    if cfg["variant"] == "cait":
        return [("patch_transformer", cfg["depth"]), ("cls_transformer", cfg["cls_depth"])]
    return [("transformer", cfg["depth"])]


def _parity(compute, names):
    import torch
    from oracle import ref_numpy, ref_torch, spec
    from util import oracle_cfg, rand_images
    for name, b in names:
        try:
            cfg = oracle_cfg(name)
            P = spec.init_params(cfg, 1, randomize_all=True)
            m = _model(name, compute, b, P)
            img = rand_images(cfg, b)
            t0 = time.time()
            logits = m(img, training=False)
            t1 = time.time()
            ref = ref_numpy.forward(cfg, P, img)
            err = np.abs(logits - ref).max()
            extra = ""
            if compute == "bf16":
                emu = ref_torch.forward(cfg, ref_torch.to_torch(P), torch.tensor(img, dtype=torch.float64), q=ref_torch.bf16_round).numpy()
                extra = f" vs bf16-oracle {np.abs(logits - emu).max():.3e} (oracle/oracle {np.abs(emu - ref).max():.3e})"
            print(f"[{compute}] {name} b={b}: max|dlogit| {err:.3e}{extra} finite={np.isfinite(logits).all()} ({t1 - t0:.2f}s)", flush=True)
            if not (err < (1e-3 if compute == "fp32" else 5e-2)):
                _bisect(m, cfg, P, img, _prefixes(cfg))
        except Exception:
            print(f"[{compute}] {name}: EXCEPTION")
            traceback.print_exc()


def stage_parity_fp32():
    _parity("fp32", [("vit_small", 3), ("vit_rect_mean", 2), ("vit_noproj", 2), ("deepvit_small", 2), ("cait_small", 2), ("cfg1_readme", 1)])


def stage_parity_bf16():
    _parity("bf16", [("vit_bf16_small", 3), ("deepvit_bf16_small", 2), ("cait_bf16_small", 2), ("cfg1_readme", 2), ("cfg2_vit_b16", 2)])
    for env in ({"VITX_GENERIC_ATTN": "1"}, {"VITX_GENERIC_GEMM": "1"}, {"VITX_GENERIC_ATTN": "1", "VITX_GENERIC_GEMM": "1"},
                {"VITX_GEMM_KERNEL": "1"}, {"VITX_GEMM_KERNEL": "3"}):
        os.environ.update(env)
        print("== with", env)
        _parity("bf16", [("vit_bf16_small", 3), ("cfg2_vit_b16", 2)])
        for k in env:
            os.environ.pop(k)


def stage_grads():
    from oracle import ref_torch, spec
    from util import oracle_cfg, rand_images, rel_max_err
    for compute, names in (("fp32", ["vit_small", "vit_rect_mean", "vit_noproj", "deepvit_small", "cait_small"]),
                           ("bf16", ["vit_bf16_small", "deepvit_bf16_small", "cait_bf16_small"])):
        for name in names:
            try:
                cfg = oracle_cfg(name)
                P = spec.init_params(cfg, 1, randomize_all=True)
                m = _model(name, compute, 2, P)
                img = rand_images(cfg, 2)
                dl = (np.random.default_rng(5).standard_normal((2, cfg["num_classes"])) / 2).astype(np.float32)
                m(img, training=False)
                grads, dimg = m.backward(dl, want_dimg=True)
                _, ref, rdimg = ref_torch.forward_backward(cfg, P, img, dl, want_dimg=True)
                errs = sorted(((rel_max_err(grads[k], ref[k]), k) for k in ref), reverse=True)
                print(f"[{compute}] {name}: worst grad rel errs: " + ", ".join(f"{k} {e:.2e}" for e, k in errs[:4]) +
                      f" | dimg {rel_max_err(dimg, rdimg):.2e}", flush=True)
                if errs[0][0] > (1e-3 if compute == "fp32" else 6e-2):
                    for e, k in errs:
                        print(f"      {k:50s} {e:.3e}")
            except Exception:
                print(f"[{compute}] {name}: EXCEPTION")
                traceback.print_exc()
    # fused-attention / MFMA-GEMM switches on the bf16 gradient path
    for env in ({"VITX_GENERIC_ATTN": "1"}, {"VITX_GENERIC_GEMM": "1"}):
        os.environ.update(env)
        try:
            name = "vit_bf16_small"
            cfg = oracle_cfg(name)
            P = spec.init_params(cfg, 1, randomize_all=True)
            m = _model(name, "bf16", 2, P)
            img = rand_images(cfg, 2)
            dl = (np.random.default_rng(5).standard_normal((2, cfg["num_classes"])) / 2).astype(np.float32)
            m(img, training=False)
            grads, _ = m.backward(dl)
            _, ref, _ = ref_torch.forward_backward(cfg, P, img, dl)
            errs = sorted(((rel_max_err(grads[k], ref[k]), k) for k in ref), reverse=True)
            print(f"[bf16 {env}] {name}: worst grad rel errs: " + ", ".join(f"{k} {e:.2e}" for e, k in errs[:4]), flush=True)
        except Exception:
            traceback.print_exc()
        for k in env:
            os.environ.pop(k)


def stage_bench_quick():
    for extra in ([], ["--workload", "vit_readme_256", "--batch", "64"]):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", *extra]
        print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        print(r.stdout[-6000:])
        print(r.stderr[-3000:])


if __name__ == "__main__":
    stage = sys.argv[1]
    t0 = time.time()
    print(f"######## stage {stage}", flush=True)
    try:
        globals()["stage_" + stage]()
    except Exception:
        traceback.print_exc()
    print(f"######## stage {stage} done in {time.time() - t0:.1f}s", flush=True)
