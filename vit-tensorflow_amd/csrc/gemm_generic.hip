// Generic strided / batched GEMM with fp32 FMA accumulation (k-ordered chain, i.e. exact fp32
// arithmetic): the FP32_PARITY compute mode's Dense layers (vit.py:39,42,59,63,143,156), and the
// materialised attention path (scores / AV and their VJPs: vit.py:77,81; deepvit.py:79,87;
// cait.py:121,127) used by parity mode and by the DeepViT / CaiT variants.
//   C[z][m][n] = alpha * sum_k A[z](m,k) * B[z](k,n)   (+ epilogue)
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]: any transpose is a stride choice.
// z = zb*nh + zh with independent (batch, head) strides so that [b, n, (h d)] tensors are addressed
// in place (the reference's 'b n (h d) -> b h n d' rearranges, vit.py:74,82, are pure addressing).
#include "kernels.h"

namespace {

constexpr int GT = 64;   // tile M = tile N
constexpr int GK = 16;   // tile K

template <typename TA, typename TB, int MODE, typename TO>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GenericGemmArgs g, EpiParams ep) {
  __shared__ float As[GK][GT + 4];
  __shared__ float Bs[GK][GT + 4];
  const int z = blockIdx.z, zb = z / g.nh, zh = z - zb * g.nh;
  const TA* A = (const TA*)g.A + (int64_t)zb * g.sAb + (int64_t)zh * g.sAh;
  const TB* B = (const TB*)g.B + (int64_t)zb * g.sBb + (int64_t)zh * g.sBh;
  const int64_t out_off = (int64_t)zb * ep.out_batch_stride + (int64_t)zh * ep.out_head_stride;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int t = threadIdx.x;
  const int tm = (t / 16) * 4, tn = (t % 16) * 4;
  float acc[4][4] = {};

  for (int k0 = 0; k0 < g.K; k0 += GK) {
    // stage A tile [GT m][GK k] -> As[k][m]; thread mapping follows the unit-stride axis
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int mm, kk;
      if (g.sak == 1) { kk = t % GK; mm = t / GK + 16 * i; } else { mm = t % GT; kk = t / GT + 4 * i; }
      const int gm = m0 + mm, gk = k0 + kk;
      float v = 0.f;
      if (gm < g.M && gk < g.K) v = ldf<TA>(A + (int64_t)gm * g.sam + (int64_t)gk * g.sak);
      As[kk][mm] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int nn, kk;
      if (g.sbk == 1) { kk = t % GK; nn = t / GK + 16 * i; } else { nn = t % GT; kk = t / GT + 4 * i; }
      const int gn = n0 + nn, gk = k0 + kk;
      float v = 0.f;
      if (gn < g.N && gk < g.K) v = ldf<TB>(B + (int64_t)gk * g.sbk + (int64_t)gn * g.sbn);
      Bs[kk][nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; ++kk) {
      const float4 a = *(const float4*)&As[kk][tm];
      const float4 b = *(const float4*)&Bs[kk][tn];
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + tm + i;
    epilogue_apply4<MODE, TO>(ep, row, n0 + tn, make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]), out_off);
  }
}

template <typename TA, typename TB, typename TO>
void launch_mode(const GenericGemmArgs& g, const EpiParams& ep, int mode, int m_tiles_rows, hipStream_t s) {
  dim3 grid((unsigned)ceil_div(g.N, GT), (unsigned)ceil_div(m_tiles_rows, GT), (unsigned)(g.nb * g.nh));
  dim3 block(256);
#define VITX_CASE(MODE) case MODE: hipLaunchKernelGGL((gemm_generic_kernel<TA, TB, MODE, TO>), grid, block, 0, s, g, ep); break;
  switch (mode) {
    VITX_CASE(EPI_STORE) VITX_CASE(EPI_STORE_F32) VITX_CASE(EPI_BIAS_GELU) VITX_CASE(EPI_BIAS_RESID)
    VITX_CASE(EPI_PATCH) VITX_CASE(EPI_GELU_BWD) VITX_CASE(EPI_PARTIAL)
    default: break;
  }
#undef VITX_CASE
}

}  // namespace

// ta/tb/to: 0 = fp32, 1 = bf16.  Rows up to round_up(M, 64) are visited when ep.zero_pad is set
// (the caller guarantees the output buffers are row-padded to a multiple of 256).
void launch_gemm_generic(const GenericGemmArgs& g, const EpiParams& ep, int mode, int ta, int tb, int to, hipStream_t s) {
  if (g.x3 && gemm_bf16x3_supported(g, ta, tb, to)) { launch_gemm_bf16x3(g, ep, mode, s); return; }   // BF16X3 mode: split operands, bf16 matrix pipe
  if (gemm_f32_mfma_supported(g, ta, tb, to)) { launch_gemm_f32_mfma(g, ep, mode, s); return; }   // fp32 matrix pipe, same bits
  const int rows = g.M;
  if (ta == 0 && tb == 0 && to == 0) launch_mode<float, float, float>(g, ep, mode, rows, s);
  else if (ta == 1 && tb == 1 && to == 1) launch_mode<bf16_t, bf16_t, bf16_t>(g, ep, mode, rows, s);
  else if (ta == 0 && tb == 1 && to == 1) launch_mode<float, bf16_t, bf16_t>(g, ep, mode, rows, s);
  else if (ta == 1 && tb == 0 && to == 1) launch_mode<bf16_t, float, bf16_t>(g, ep, mode, rows, s);
  else if (ta == 1 && tb == 1 && to == 0) launch_mode<bf16_t, bf16_t, float>(g, ep, mode, rows, s);
  else if (ta == 0 && tb == 1 && to == 0) launch_mode<float, bf16_t, float>(g, ep, mode, rows, s);
  else if (ta == 1 && tb == 0 && to == 0) launch_mode<bf16_t, float, float>(g, ep, mode, rows, s);
  else launch_mode<float, float, bf16_t>(g, ep, mode, rows, s);
}
