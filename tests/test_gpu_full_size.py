"""GPU tier (-m gpu): the BENCHMARKED shapes themselves against an oracle ("the benchmarked shape itself never meets an oracle").

1. Every Dense launch of the ViT-B/16 step at batch 256 (M = 256 x 197 = 50432 token rows; vit.py:39,42,59,63 and their VJPs): the MFMA
   kernels -- each persistent tile variant the per-shape measurement can pick, each fused epilogue, the 7-28-slice split-K weight gradient --
   against the k-ordered fp32-FMA kernel on the same bf16 operands, compared on the device (vitx_check_gemm).
2. The whole step at the benchmark configuration (ViT-B/16 224, depth 12, batch 256, bf16 mode): logits and every parameter gradient
   against oracle/ref_torch.py evaluated in fp32 ON THE GPU BOX'S TORCH (the checker may use torch; the product does not).
3. ViT-L/16 widths are pinned by tests/golden/ref_vit_l16_depth2.npz (test_gpu_ref_fixtures.py)."""
import ctypes as C

import numpy as np
import pytest

from oracle import ref_torch, spec
from util import gate, make_engine_model, oracle_cfg, rand_images

pytestmark = pytest.mark.gpu
M_TOKENS = 256 * 197

# (N, K, epilogues): 1 = bias + fp32 residual, 2 = bias + GELU (act and gelu' in bf16), 3 = bf16 store, 4 = x stored gelu' + fused column sums
NT_LAUNCHES = [(3072, 768, (2, 4, 3)), (768, 3072, (1, 3)), (768, 768, (1, 3)), (2304, 768, (3,)), (768, 2304, (3,))]
NT_VARIANTS = (6, 7, 9, 10, 11, 13)   # the persistent variants: lockstep 256 / 320 rows (fall-back beyond 2 GiB operands), pipelined + wave-private epilogue 320 / 256 rows,
                                    # 9 / 10: the pipelined kernel with 192 x 128 tiles on 4 waves, two workgroups per CU (persistent / one tile per workgroup)
BF16_OUT_TOL = 1.1e-2    # ~2x observed (5.2e-3 = one bf16 ulp): the two kernels add in different orders, a value on a rounding boundary flips one ulp
GELU_OUT2_TOL = 1.5e-2   # gelu'(h) / gelu(h) of a pre-activation that flipped (observed 7.2e-3)
F32_OUT_TOL = 1e-3
COLSUM_TOL = 8e-3          # observed 3.8e-3


def _handle():
    from vit_tensorflow import _native as N
    m = make_engine_model("vit_bf16_small", "bf16", 1)
    m.build((1,))
    return N, m


@pytest.mark.parametrize("N_, K, epis", NT_LAUNCHES)
def test_dense_launches_of_the_benchmarked_step_match_the_fp32_fma_kernel(N_, K, epis):
    N, m = _handle()
    errs = (C.c_float * 2)()
    for epi in epis:
        for kern in NT_VARIANTS:
            N.check(N.lib().vitx_check_gemm(m._handle, 0, M_TOKENS, N_, K, kern, epi, errs))
            what = f"M {M_TOKENS} N {N_} K {K} epilogue {epi} variant {kern}"
            gate(errs[0], F32_OUT_TOL if epi in (0, 1) else BF16_OUT_TOL, what, f"gemm_full_size_epi{epi}")
            if epi == 2:
                gate(errs[1], GELU_OUT2_TOL, what + " (second output)", "gemm_full_size_epi2_out2")
            if epi == 4:
                gate(errs[1], COLSUM_TOL, what + " (fused column sums)", "gemm_full_size_colsum")


@pytest.mark.parametrize("in_, out", [(768, 3072), (3072, 768), (768, 768), (768, 2304), (1024, 4096)])
def test_weight_gradient_launches_at_full_token_count_match_the_fp32_fma_kernel(in_, out):
    N, m = _handle()
    errs = (C.c_float * 2)()
    N.check(N.lib().vitx_check_gemm(m._handle, 1, in_, out, M_TOKENS, 0, 0, errs))
    assert errs[1] >= 2, errs[1]                      # really a split-K launch (7 .. 28 slices at these shapes)
    # 50432-term fp32 sums in two different orders: |dW| ~ sqrt(50432) / 16 ~ 14, relative rounding ~ 1e-6 sqrt(788)
    gate(errs[0], 1e-3, f"dW {in_} x {out} over {M_TOKENS} token rows, {int(errs[1])} slices", "gemm_tn_full_size")
    print(f"[tn] {in_}x{out}: {int(errs[1])} K slices, max rel diff {errs[0]:.3e}")


def test_benchmark_configuration_vit_b16_batch_256_bf16_against_the_torch_oracle_on_the_gpu():
    """BASELINE.json configs[1] exactly as bench.py runs it (depth 12, batch 256, bf16 mode), against the fp32 evaluation of the oracle's
    restatement of vit.py:159-177 + autograd on the same weights and images.  Gates = 2x what MI355X produced (printed by the gate recorder)."""
    import torch
    name, b = "cfg2_vit_b16", 256
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, seed=3, randomize_all=True)
    m = make_engine_model(name, "bf16", b, P)
    img = rand_images(cfg, b, 11)
    dl = (np.random.default_rng(12).standard_normal((b, cfg["num_classes"])) / b).astype(np.float32)
    logits = m(img, training=True)
    grads, dimg = m.backward(dl, want_dimg=True)

    dev = torch.device("cuda:0")
    Pt = {k: torch.tensor(np.asarray(v, np.float32), device=dev, requires_grad=True) for k, v in P.items()}
    x = torch.tensor(img, device=dev, requires_grad=True)
    ref = ref_torch.forward(cfg, Pt, x)
    ref.backward(torch.tensor(dl, device=dev))
    ref_logits = ref.detach().double().cpu().numpy()
    std = float(ref_logits.std())
    e_logit = float(np.abs(logits - ref_logits).max()) / max(1.0, std)
    print(f"[full-size] logits: max|d| / std = {e_logit:.3e} (std {std:.3f})")
    gate(e_logit, 4.1e-2, "logits, 12 layers of bf16 operands", "full_size_logits")          # observed 2.02e-2
    worst = ("", 0.0)
    for n, _, _ in spec.param_spec(cfg):
        r = Pt[n].grad.detach().double().cpu().numpy()
        g = np.asarray(grads[n], np.float64)
        scale = float(np.abs(r).max()) + 1e-30
        e_max = float(np.abs(g - r).max()) / scale
        e_sum = abs(float(g.sum()) - float(r.sum())) / (float(np.abs(r).sum()) + 1e-30)
        e_l2 = float(np.linalg.norm(g - r) / (np.linalg.norm(r) + 1e-30))
        worst = max(worst, (n, e_max), key=lambda t: t[1])
        gate(e_max, 2.2e-2, f"grad {n} (max error / max)", "full_size_grad_max")            # observed 1.08e-2
        gate(e_sum, 2.0e-3, f"grad {n} (sum / abs-sum)", "full_size_grad_sum")              # observed 6.4e-4
        gate(e_l2, 1.6e-2, f"grad {n} (relative L2)", "full_size_grad_l2")                  # observed 7.8e-3
    rd = x.grad.detach().double().cpu().numpy()
    gate(float(np.abs(dimg - rd).max()) / (float(np.abs(rd).max()) + 1e-30), 1.5e-2, "d(img)", "full_size_dimg")   # observed 7.3e-3
    print(f"[full-size] worst gradient: {worst[0]} {worst[1]:.3e}")


def test_vit_b16_depth_12_batch_64_bf16x3_meets_the_fp32_gates_against_the_torch_oracle_on_the_gpu():
    """The BF16X3 mode (fp32 data path, split-operand bf16 MFMA GEMMs) at the full ViT-B/16 depth and a batch that fills the split kernel's
    tiles (12608 token rows, 99 row tiles; split-K weight gradients): north_star's 1e-3 on the logits, 1e-3 of each tensor's max on the gradients."""
    import torch
    name, b = "cfg2_vit_b16", 64
    cfg = oracle_cfg(name)
    P = spec.init_params(cfg, seed=3, randomize_all=True)
    m = make_engine_model(name, "bf16x3", b, P)
    img = rand_images(cfg, b, 11)
    dl = (np.random.default_rng(12).standard_normal((b, cfg["num_classes"])) / b).astype(np.float32)
    logits = m(img, training=True)
    grads, dimg = m.backward(dl, want_dimg=True)

    dev = torch.device("cuda:0")
    Pt = {k: torch.tensor(np.asarray(v, np.float32), device=dev, requires_grad=True) for k, v in P.items()}
    x = torch.tensor(img, device=dev, requires_grad=True)
    ref = ref_torch.forward(cfg, Pt, x)
    ref.backward(torch.tensor(dl, device=dev))
    ref_logits = ref.detach().double().cpu().numpy()
    e_logit = float(np.abs(logits - ref_logits).max())
    print(f"[full-size bf16x3] logits: max|d| = {e_logit:.3e} (std {float(ref_logits.std()):.3f})")
    gate(e_logit, 1e-3, "logits, 12 layers of split-operand GEMMs", "x3_full_size_logits")
    worst = ("", 0.0)
    for n, _, _ in spec.param_spec(cfg):
        r = Pt[n].grad.detach().double().cpu().numpy()
        g = np.asarray(grads[n], np.float64)
        e_max = float(np.abs(g - r).max()) / (float(np.abs(r).max()) + 1e-30)
        worst = max(worst, (n, e_max), key=lambda t: t[1])
        gate(e_max, 1e-3, f"grad {n} (max error / max)", "x3_full_size_grad_max")
    rd = x.grad.detach().double().cpu().numpy()
    gate(float(np.abs(dimg - rd).max()) / (float(np.abs(rd).max()) + 1e-30), 1e-3, "d(img)", "x3_full_size_dimg")
    print(f"[full-size bf16x3] worst gradient: {worst[0]} {worst[1]:.3e}")
