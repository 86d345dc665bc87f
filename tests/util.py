"""Shared helpers for the parity tests: build an engine model and the oracle on identical weights."""
from __future__ import annotations

import numpy as np

from oracle import spec

CONFIGS = {
    # small shapes the fp64 oracle finishes in well under a second
    "vit_small": ("vit", dict(image_size=64, patch_size=16, num_classes=10, dim=64, depth=2, heads=2, mlp_dim=128, dim_head=32)),
    "vit_rect_mean": ("vit", dict(image_size=(64, 32), patch_size=(16, 8), num_classes=12, dim=64, depth=2, heads=4, mlp_dim=192, dim_head=16, pool="mean")),
    "vit_noproj": ("vit", dict(image_size=32, patch_size=8, num_classes=7, dim=64, depth=2, heads=1, mlp_dim=128, dim_head=64)),
    "vit_bf16_small": ("vit", dict(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256, dim_head=64)),
    # more than 64 keys per row: the multi-sweep head-axis kernels (DeepViT 82 tokens; CaiT 81 patch keys / 82 class-attention keys)
    "deepvit_82tok": ("deepvit", dict(image_size=144, patch_size=16, num_classes=10, dim=64, depth=2, heads=4, mlp_dim=128, dim_head=16)),
    "cait_82tok": ("cait", dict(image_size=144, patch_size=16, num_classes=10, dim=64, depth=1, cls_depth=1, heads=4, mlp_dim=128, dim_head=16)),
    "deepvit_small": ("deepvit", dict(image_size=64, patch_size=16, num_classes=10, dim=64, depth=2, heads=4, mlp_dim=128, dim_head=16)),
    "deepvit_bf16_small": ("deepvit", dict(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, heads=4, mlp_dim=256, dim_head=32)),
    "cait_small": ("cait", dict(image_size=64, patch_size=16, num_classes=10, dim=64, depth=2, cls_depth=2, heads=4, mlp_dim=128, dim_head=16)),
    "cait_bf16_small": ("cait", dict(image_size=64, patch_size=16, num_classes=10, dim=128, depth=2, cls_depth=2, heads=4, mlp_dim=256, dim_head=32)),
    # BASELINE.json configs (reduced batch for the oracle)
    "cfg1_readme": ("vit", dict(image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=6, heads=16, mlp_dim=2048)),
    "cfg2_vit_b16": ("vit", dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072)),
    "cfg4_deepvit": ("deepvit", dict(image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=12, heads=16, mlp_dim=2048)),
    "cfg5_cait": ("cait", dict(image_size=256, patch_size=32, num_classes=1000, dim=1024, depth=24, cls_depth=2, heads=16, mlp_dim=2048)),
}


def oracle_cfg(name: str) -> dict:
    v, kw = CONFIGS[name]
    return spec.make_config(v, **kw)


def make_engine_model(name: str, compute: str = "fp32", max_batch: int = 4, params=None):
    v, kw = CONFIGS[name]
    if v == "vit":
        from vit_tensorflow import ViT as cls
    elif v == "deepvit":
        from vit_tensorflow.deepvit import DeepViT as cls
    else:
        from vit_tensorflow.cait import CaiT as cls
    m = cls(**kw, compute=compute, max_batch=max_batch, seed=0)
    if params is not None:
        m.load_state_dict({k: np.asarray(a, dtype=np.float32) for k, a in params.items()})
    return m


def rand_images(cfg: dict, b: int, seed: int = 0, hw=None) -> np.ndarray:
    h, w = hw or cfg["image_size"]
    return np.random.Generator(np.random.PCG64(seed)).standard_normal((b, h, w, 3)).astype(np.float32)


def rel_max_err(a: np.ndarray, ref: np.ndarray) -> float:
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / (np.abs(ref).max() + 1e-30))


_OBSERVED = {}


def gate(err: float, tol: float, what: str, group: str = "") -> None:
    """assert err <= tol, and remember the worst err / tol ratio per group: `report_gates()` prints them (pytest -rA shows the
    output), which is how the bf16 gates of the GPU tier are kept at about twice what MI355X actually produces."""
    if tol >= 5e-3:
        w = _OBSERVED.setdefault(group or what, (0.0, tol, what))
        if err > w[0]:
            _OBSERVED[group or what] = (float(err), tol, what)
    assert err <= tol, f"{what}: {err:.3e} > {tol:.1e}"


def report_gates() -> None:
    for k, (e, t, what) in sorted(_OBSERVED.items()):
        print(f"[gate] {k}: worst observed {e:.3e} (gate {t:.1e}, at {what})")
    _OBSERVED.clear()


def fake_rccl_lib() -> str:
    """tests/fake_rccl/libfake_rccl.so (test infrastructure: ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy over a POSIX
    shared-memory segment, stream-ordered with hipLaunchHostFunc), compiled on first use; point csrc/comm.hip at it with VITX_RCCL_LIB."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl")
    src, out = os.path.join(here, "fake_rccl.cpp"), os.path.join(here, "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", src, "-o", out, "-lrt"])
    return out
