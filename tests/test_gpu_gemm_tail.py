"""GPU tier: tail balancing of the persistent Dense launches (round 6; gemm_bf16.hip, dispatch_gemm_bf16_tail; vit.py:39,42,59,63 and their VJPs).

A launch whose tile count is not a multiple of the 256 persistent workgroups goes out as the rows of its whole rounds (256 x 256 tiles) plus the rows of
the last partial round as a second launch of small tiles (192 x 128, 128 x 128 or 256 x 128, one per workgroup).  A sub-launch is the same GEMM on a row
range with the same fused epilogue, and every output element is still accumulated by one wave in the same K order, so:
  (the fc1 launch, bias + GELU, is never split: only the 256-row tile has LDS to spare for the GELU table, the small tiles would round gelu differently)
  1. at the benchmarked row count every (epilogue, tail variant) pair matches the k-ordered fp32-FMA kernel on the same operands (vitx_check_gemm);
  2. a whole ViT-B/16 block, forward + backward, gives the SAME BITS with and without the split -- logits, d(img) and every gradient except the fc1 bias
     gradient, whose fused per-tile column sums are added over other row groups (a fixed-order fp32 sum in another order: compared to 1e-5)."""
import ctypes as C

import numpy as np
import pytest

from oracle import spec
from util import gate, make_engine_model, rand_images

pytestmark = pytest.mark.gpu
M_TOKENS = 256 * 197
TAILS = (10, 1, 3)
NT_LAUNCHES = [(3072, 768, (4, 3)), (768, 3072, (1, 3)), (768, 768, (1, 3)), (2304, 768, (3,)), (1024, 1024, (1, 3))]


@pytest.mark.parametrize("N_, K, epis", NT_LAUNCHES)
def test_tail_balanced_launches_match_the_fp32_fma_kernel_at_the_benchmarked_row_count(N_, K, epis):
    from vit_tensorflow import _native as N
    m = make_engine_model("vit_bf16_small", "bf16", 1)
    m.build((1,))
    errs = (C.c_float * 2)()
    for epi in epis:
        for t in TAILS:
            N.check(N.lib().vitx_check_gemm(m._handle, 0, M_TOKENS, N_, K, 13 | (t << 10), epi, errs))
            what = f"M {M_TOKENS} N {N_} K {K} epilogue {epi} variant 13 + tail {t}"
            gate(errs[0], 1e-3 if epi == 1 else 1.1e-2, what, f"gemm_tail_epi{epi}")
            if epi == 4:
                gate(errs[1], 8e-3, what + " (fused column sums)", "gemm_tail_colsum")


def _block(monkeypatch, tail):
    monkeypatch.setenv("VITX_GEMM_KERNEL", "13")
    monkeypatch.setenv("VITX_GEMM_TAIL_KERNEL", str(tail))
    from vit_tensorflow import ViT
    kw = dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=1, heads=12, mlp_dim=3072)
    cfg = spec.make_config("vit", **kw)
    P = spec.init_params(cfg, seed=5, randomize_all=True)
    b = 256
    m = ViT(**kw, compute="bf16", max_batch=b, seed=0)
    m.load_state_dict({k: np.asarray(v, np.float32) for k, v in P.items()})
    img = rand_images(cfg, b, 7)
    dl = (np.random.default_rng(8).standard_normal((b, 1000)) / b).astype(np.float32)
    logits = np.array(m(img, training=True), copy=True)
    grads, dimg = m.backward(dl, want_dimg=True)
    return logits, {k: np.array(v, copy=True) for k, v in grads.items()}, np.array(dimg, copy=True)


@pytest.mark.parametrize("tail", TAILS)
def test_a_vit_b16_block_gives_the_same_bits_with_and_without_the_tail_launch(tail, monkeypatch):
    l0, g0, d0 = _block(monkeypatch, 0)
    l1, g1, d1 = _block(monkeypatch, tail)
    assert np.array_equal(l0, l1)
    assert np.array_equal(d0, d1)
    for k in g0:
        if k.endswith("mlp.fc1.bias"):
            e = float(np.abs(g0[k] - g1[k]).max() / (np.abs(g0[k]).max() + 1e-30))
            assert e <= 1e-5, (k, e)
        else:
            assert np.array_equal(g0[k], g1[k]), k
