// fp32 GEMM on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) for the FP32_PARITY compute mode: the Dense layers
// (vit.py:39,42,59,63,143,156), their VJPs, and the materialised attention products (vit.py:77,81; deepvit.py:79,87; cait.py:121,127).
//   C[z][m][n] = alpha * sum_k A[z](m,k) * B[z](k,n)   (+ the shared fused epilogues)
// Same interface as the scalar-FMA kernel of gemm_generic.hip (any transpose is a stride choice; (batch, head) strides address
// [b, n, (h d)] tensors in place) and the SAME arithmetic: the fp32 MFMA is, bit for bit, a k-ordered fmaf chain
// D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) with one rounding per product (MI355X guide, "FP32-input MFMA"), which is exactly the
// chain the scalar kernel evaluates -- so the two kernels return identical bits (tests/test_gpu_edges.py checks that) and the
// 1e-3 parity gate is met at matrix-pipe speed: 64 FLOP/clk/SIMD = 157 TFLOP/s peak, 1/16 of the bf16 rate, ~3x the VALU kernel.
//
// gfx950 mapping: 128 x 128 x 16 tiles, 4 waves (2 x 2), 64 x 64 per wave = 2 x 2 MFMA tiles (64 accumulator registers).
// Operands are staged global -> registers -> LDS as [k][m] / [k][n] planes (the unit-stride axis of the source decides the thread
// mapping, 16-B loads where the layout allows), so an MFMA operand read is one conflict-free ds_read_b32 per lane:
// lane (i = lane & 31, k = lane >> 5) takes plane k, column i.  Operands are swapped (D^T = B^T A^T) so that a lane ends up with
// four consecutive output columns of one row, the shape every fused epilogue consumes.  LDS is double buffered; the global loads
// of K-tile t+1 are in flight while tile t is multiplied.
#include "kernels.h"

namespace {

constexpr int FM = 128, FN = 128, FK = 16, FPAD = 4;

template <int MODE>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(GenericGemmArgs g, EpiParams ep, int a_mode, int b_mode) {
  __shared__ float As[2][FK][FM + FPAD];
  __shared__ float Bs[2][FK][FN + FPAD];
  const int z = blockIdx.z, zb = z / g.nh, zh = z - zb * g.nh;
  const float* A = (const float*)g.A + (int64_t)zb * g.sAb + (int64_t)zh * g.sAh;
  const float* B = (const float*)g.B + (int64_t)zb * g.sBb + (int64_t)zh * g.sBh;
  const int64_t out_off = (int64_t)zb * ep.out_batch_stride + (int64_t)zh * ep.out_head_stride;
  const int m0 = blockIdx.y * FM, n0 = blockIdx.x * FN;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // staging: 2048 elements per operand tile, 8 per thread.  mode 1: 16-B loads along k (source k-contiguous), mode 2: 16-B loads
  // along the row/column index (source m- / n-contiguous), mode 0: scalar loads, thread index along the unit(-ish) stride
  float ra[8], rb[8];
  auto load_tile = [&](const float* P, int mode, int64_t s_idx, int64_t s_k, int idx0, int lim, int k0, float (&r)[8]) {
    if (mode == 1) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int idx = p * 64 + (t >> 2), kk = (t & 3) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx0 + idx < lim && k0 + kk < g.K) v = *(const float4*)(P + (int64_t)(idx0 + idx) * s_idx + (k0 + kk));
        r[p * 4 + 0] = v.x; r[p * 4 + 1] = v.y; r[p * 4 + 2] = v.z; r[p * 4 + 3] = v.w;
      }
    } else if (mode == 2) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int kk = p * 8 + (t >> 5), idx = (t & 31) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx0 + idx < lim && k0 + kk < g.K) v = *(const float4*)(P + (int64_t)(k0 + kk) * s_k + (idx0 + idx));
        r[p * 4 + 0] = v.x; r[p * 4 + 1] = v.y; r[p * 4 + 2] = v.z; r[p * 4 + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        int idx, kk;
        if (s_k == 1) { kk = t & 15; idx = (t >> 4) + 16 * p; } else { idx = t & 127; kk = (t >> 7) + 2 * p; }
        float v = 0.f;
        if (idx0 + idx < lim && k0 + kk < g.K) v = P[(int64_t)(idx0 + idx) * s_idx + (int64_t)(k0 + kk) * s_k];
        r[p] = v;
      }
    }
  };
  auto store_tile = [&](float (*S)[FM + FPAD], int mode, int64_t s_k, const float (&r)[8]) {
    if (mode == 1) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int idx = p * 64 + (t >> 2), kk = (t & 3) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) S[kk + e][idx] = r[p * 4 + e];
      }
    } else if (mode == 2) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int kk = p * 8 + (t >> 5), idx = (t & 31) * 4;
        *(float4*)&S[kk][idx] = make_float4(r[p * 4 + 0], r[p * 4 + 1], r[p * 4 + 2], r[p * 4 + 3]);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        int idx, kk;
        if (s_k == 1) { kk = t & 15; idx = (t >> 4) + 16 * p; } else { idx = t & 127; kk = (t >> 7) + 2 * p; }
        S[kk][idx] = r[p];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (g.K + FK - 1) / FK;
  if (nkt > 0) {
    load_tile(A, a_mode, g.sam, g.sak, m0, g.M, 0, ra);
    load_tile(B, b_mode, g.sbn, g.sbk, n0, g.N, 0, rb);
  }
  const int col = lane & 31, kh = lane >> 5;
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    store_tile(As[buf], a_mode, g.sak, ra);
    store_tile(Bs[buf], b_mode, g.sbk, rb);
    __syncthreads();           // tile kt visible; every wave has finished reading buffer buf (last used by tile kt-2) one barrier ago
    if (kt + 1 < nkt) {
      load_tile(A, a_mode, g.sam, g.sak, m0, g.M, (kt + 1) * FK, ra);
      load_tile(B, b_mode, g.sbn, g.sbk, n0, g.N, (kt + 1) * FK, rb);
    }
#pragma unroll
    for (int kp = 0; kp < FK / 2; ++kp) {
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = As[buf][2 * kp + kh][wm * 64 + i * 32 + col];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = Bs[buf][2 * kp + kh][wn * 64 + j * 32 + col];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  // lane: output row m = lane & 31 of each 32 x 32 tile, columns 8q + 4 (lane >> 5) + {0..3}
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wm * 64 + i * 32 + col;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        epilogue_apply4<MODE, float>(ep, row, n0 + wn * 64 + j * 32 + 8 * q + 4 * kh,
                                     make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]), out_off);
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// fp32 operands and outputs only; problems too small to fill a 128 x 128 tile stay on the 64 x 64 scalar kernel
static int g_f32_mfma_on = 1;
void gemm_f32_mfma_read_env() {   // called by engine_create: the A/B test flips the variable between two handles of one process
  const char* v = vitx_env("VITX_F32_MFMA");
  g_f32_mfma_on = v ? atoi(v) : 1;
}
bool gemm_f32_mfma_supported(const GenericGemmArgs& g, int ta, int tb, int to) {
  const int on = g_f32_mfma_on;
  return on && ta == 0 && tb == 0 && to == 0 && g.M >= 64 && g.N >= 64 && g.K >= 8;
}

void launch_gemm_f32_mfma(const GenericGemmArgs& g, const EpiParams& ep, int mode, hipStream_t s) {
  // 16-B loads along k need k-contiguous rows that start 16-B aligned for every (batch, head); likewise along m / n
  auto pick = [&](const void* P, int64_t s_idx, int64_t s_k, int extent, int64_t sb, int64_t sh) {
    const bool strides_ok = sb % 4 == 0 && sh % 4 == 0 && al16(P);
    if (s_k == 1 && s_idx % 4 == 0 && g.K % 4 == 0 && strides_ok) return 1;
    if (s_idx == 1 && s_k % 4 == 0 && extent % 4 == 0 && strides_ok) return 2;
    return 0;
  };
  const int a_mode = pick(g.A, g.sam, g.sak, g.M, g.sAb, g.sAh);
  const int b_mode = pick(g.B, g.sbn, g.sbk, g.N, g.sBb, g.sBh);
  dim3 grid((unsigned)ceil_div(g.N, FN), (unsigned)ceil_div(g.M, FM), (unsigned)(g.nb * g.nh)), block(256);
#define VITX_CASE(MODE) case MODE: hipLaunchKernelGGL((gemm_f32_mfma_kernel<MODE>), grid, block, 0, s, g, ep, a_mode, b_mode); break;
  switch (mode) {
    VITX_CASE(EPI_STORE) VITX_CASE(EPI_STORE_F32) VITX_CASE(EPI_BIAS_GELU) VITX_CASE(EPI_BIAS_RESID)
    VITX_CASE(EPI_PATCH) VITX_CASE(EPI_GELU_BWD) VITX_CASE(EPI_PARTIAL)
    default: break;
  }
#undef VITX_CASE
}
