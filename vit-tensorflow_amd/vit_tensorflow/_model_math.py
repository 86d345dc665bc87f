"""Closed-form size figures of the models this package builds (no device work): algorithmic FLOPs per image and the identity of
the kernel sources a measurement belongs to.  Used by bench.py and the tools; lives in the package so that the benchmark does not
import anything from oracle/ (which is test infrastructure)."""
from __future__ import annotations

import hashlib
import os
from typing import Tuple, Union

_HERE = os.path.dirname(os.path.abspath(__file__))


def _pair(t) -> Tuple[int, int]:
    return t if isinstance(t, tuple) else (t, t)


def flops_per_image(variant: str = "vit", *, image_size: Union[int, Tuple[int, int]], patch_size: Union[int, Tuple[int, int]], num_classes: int,
                    dim: int, depth: int, heads: int, mlp_dim: int, dim_head: int = 64, channels: int = 3, num_parallel_branches: int = 1,
                    fwd_only: bool = False, **_ignored) -> float:
    """Algorithmic FLOPs of one image: multiply-add = 2, backward = 2 x forward, no recomputation counted.
        fwd = 2 Np pd d + L (2 N d 3I + 2 N^2 I + 2 N^2 I + 2 N I d + 4 N d m) + 2 d nc
    with Np patches, N tokens (Np + 1; CaiT's patch stage: Np), I = heads x dim_head, pd = p1 p2 C, L = depth (x parallel branches).
    ViT-B/16 at 224 px: 35.128 GFLOP forward, 105.383 GFLOP forward + backward."""
    ih, iw = _pair(image_size)
    ph, pw = _pair(patch_size)
    np_ = (ih // ph) * (iw // pw)
    pd = ph * pw * channels
    inner = heads * dim_head
    n = np_ if variant == "cait" else np_ + 1
    layers = depth * max(1, int(num_parallel_branches))
    fwd = 2 * np_ * pd * dim + layers * (2 * n * dim * 3 * inner + 4 * n * n * inner + 2 * n * inner * dim + 4 * n * dim * mlp_dim) + 2 * dim * num_classes
    return float(fwd if fwd_only else 3 * fwd)


def kernel_source_id() -> str:
    """Digest of the sources of the bf16 MFMA GEMM family (the kernels `roofline` is about: gemm_bf16*.hip with their epilogues and
    device helpers): profiles that quote per-build counters of that family (PMC traffic) carry it, and bench.py refuses to quote a
    profile taken on other GEMM sources.  Changes to other kernels (attention, LayerNorm, the DeepViT / CaiT paths) do not move it."""
    h = hashlib.sha1()
    root = os.path.join(_HERE, "..", "csrc")
    files = [os.path.join(root, f) for f in ("common.h", "epilogue.h", "gemm_bf16_common.h", "gemm_bf16.hip", "gemm_bf16_pipe.hip", "gemm_bf16_tn.hip", "gemm_bf16x3.hip")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    try:   # the compile flags are part of what was built (per-source scheduler strategies, -fno-honor-nans): build.py's own digest of them
        import importlib.util
        spec = importlib.util.spec_from_file_location("vitx_build_flags", os.path.join(_HERE, "..", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        h.update(b"flags\0" + mod.flags_id().encode())
    except Exception:   # a tree without build.py (pre-built library only): the sources alone
        pass
    return h.hexdigest()[:16]
