"""Calibration only (not on the product path): what the vendor library (hipBLASLt/rocBLAS through torch.matmul) reaches on this box
for the GEMM shapes of the ViT-B/16 step, same operand distribution as vitx_bench_gemm.  Puts our MFMA kernels' TFLOP/s in context
(the loop is power/clock-limited with random data)."""
import sys, time
import torch

def bench(M, N, K, iters=20, zero=False, layout="nt"):
    dev = "cuda"
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16) if layout == "nt" else torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    if zero:
        a.zero_(); b.zero_()
    f = (lambda: a @ b.t()) if layout == "nt" else (lambda: a @ b)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9

if __name__ == "__main__":
    shapes = [(50432, 768, 768), (50432, 2304, 768), (50432, 3072, 768), (50432, 768, 3072), (8192, 8192, 8192), (4096, 4096, 4096)]
    for (M, N, K) in shapes:
        for layout in ("nt", "nn"):
            ms, tf = bench(M, N, K, layout=layout)
            print(f"torch.matmul bf16 {layout} M{M} N{N} K{K}: {ms:.4f} ms {tf:.1f} TFLOP/s", flush=True)
    ms, tf = bench(8192, 8192, 8192, zero=True)
    print(f"torch.matmul bf16 nt ZERO operands 8192^3: {ms:.4f} ms {tf:.1f} TFLOP/s")
    # wgrad shape: C[768,3072] = A[50432,768]^T B[50432,3072]
    a = torch.randn(50432, 768, device="cuda", dtype=torch.bfloat16); b = torch.randn(50432, 3072, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): a.t() @ b
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): a.t() @ b
    torch.cuda.synchronize(); ms = (time.time() - t) / 20 * 1e3
    print(f"torch.matmul bf16 tn (wgrad) 768x3072x50432: {ms:.4f} ms {2.0*768*3072*50432/ms/1e9:.1f} TFLOP/s")
