// Batched small GEMM on MFMA for the MATERIALISED attention path (DeepViT Re-attention, CaiT talking heads: the head-mixing ops
// need the whole [b,h,N,N] score tensor, so the fused flash-style kernel does not apply; deepvit.py:79-87, cait.py:121-127).
//   C[b,h][M,N] = alpha * A[b,h][M,K] * B[b,h][K,N]     for M, N, K <= 128 (64/65-token configurations)
// with arbitrary element strides for A and B (the six products of attention forward+backward use q/k/v/dO head slices of packed
// bf16 rows and fp32 score matrices, plain or transposed).  One workgroup (4 waves) per (image, head):
//   * both operands are staged ONCE into LDS in canonical K-contiguous bf16 form (A as [M][K], B as [N][K]); the loader walks
//     the source along whichever index is contiguous in memory with 16-B vector loads and scatters into the canonical image,
//     converting fp32 scores to bf16 on the way; padding rows / columns are zero-filled;
//   * v_mfma_f32_16x16x32_bf16 with swapped operands, so a lane ends up with 4 consecutive output columns of one row.
// This replaces the strided fp32-FMA kernel (15-28 TFLOP/s on these shapes) in bf16 mode; parity mode keeps the fp32 kernel.
#include "kernels.h"

namespace {

constexpr int PAD = 8;   // canonical rows are K_pad + 8 bf16 long: 16-B ds_read_b128 fragments of 16 rows fall on distinct banks

template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    bf16x8 x = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
  }
};
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
};

// canonical dst[r][c] (pitch elements) = src[r*sr + c*sc] for r < R, c < Cn; zero elsewhere in [Rp][Cp]
template <typename T>
__device__ __forceinline__ void stage_canonical(const T* src, int64_t sr, int64_t sc, int R, int Cn, int Rp, int Cp, bf16_t* dst, int pitch,
                                                int tid, int nthr) {
  const bool vec_ok = (((uintptr_t)src) % 16 == 0);
  if (sc == 1 && vec_ok && (sr * (int64_t)sizeof(T)) % 16 == 0) {          // contiguous along c (the K index of this operand)
    const int cgroups = Cp / 8;
    for (int e = tid; e < Rp * cgroups; e += nthr) {
      const int r = e / cgroups, c0 = (e - r * cgroups) * 8;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (r < R && c0 < Cn) {
        if (c0 + 8 <= Cn) Vec8<T>::load(src + r * sr + c0, v);
        else for (int i = 0; i < 8 && c0 + i < Cn; ++i) v[i] = (float)src[r * sr + c0 + i];
      }
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
      *(bf16x8*)(dst + r * pitch + c0) = o;
    }
  } else if (sr == 1 && vec_ok && (sc * (int64_t)sizeof(T)) % 16 == 0) {   // contiguous along r: vector load, scatter
    const int rgroups = Rp / 8;
    for (int e = tid; e < Cp * rgroups; e += nthr) {
      const int c = e / rgroups, r0 = (e - c * rgroups) * 8;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (c < Cn && r0 < R) {
        if (r0 + 8 <= R) Vec8<T>::load(src + c * sc + r0, v);
        else for (int i = 0; i < 8 && r0 + i < R; ++i) v[i] = (float)src[c * sc + r0 + i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[(r0 + i) * pitch + c] = (bf16_t)v[i];
    }
  } else if (sr < sc) {                                                    // unaligned rows (DeepViT's 65-float score rows) with r the fast index in memory:
    for (int e = tid; e < Rp * Cp; e += nthr) {                            // lanes along r (coalesced 4-B loads; DeepViT cfg4 -0.06 ms per step, r4u)
      const int c = e / Rp, r = e - c * Rp;
      dst[r * pitch + c] = (r < R && c < Cn) ? (bf16_t)(float)src[r * sr + c * sc] : (bf16_t)0.f;
    }
  } else {                                                                 // anything else: element by element, lanes along c
    for (int e = tid; e < Rp * Cp; e += nthr) {
      const int r = e / Cp, c = e - r * Cp;
      dst[r * pitch + c] = (r < R && c < Cn) ? (bf16_t)(float)src[r * sr + c * sc] : (bf16_t)0.f;
    }
  }
}

template <typename TA, typename TO>
__device__ __forceinline__ void bgemm_mfma_body(const GenericGemmArgs& g, const EpiParams& ep, int Mp, int Np, int Kp, char* smem) {
  const int pitch = Kp + PAD;
  bf16_t* As = (bf16_t*)smem;
  bf16_t* Bs = As + Mp * pitch;
  const int bh = blockIdx.x, b = bh / g.nh, h = bh - b * g.nh;
  const TA* A = (const TA*)g.A + b * g.sAb + h * g.sAh;
  const bf16_t* B = (const bf16_t*)g.B + b * g.sBb + h * g.sBh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_canonical<TA>(A, g.sam, g.sak, g.M, g.K, Mp, Kp, As, pitch, tid, 256);
  stage_canonical<bf16_t>(B, g.sbn, g.sbk, g.N, g.K, Np, Kp, Bs, pitch, tid, 256);   // canonical B is [n][k]
  __syncthreads();

  const int mt = Mp / 16, nt = Np / 16;
  const int frow = lane & 15, fk = (lane >> 4) * 8;
  TO* out = (TO*)ep.out + b * ep.out_batch_stride + h * ep.out_head_stride;
  for (int t = wave; t < mt * nt; t += 4) {
    const int tm = t / nt, tn = t - tm * nt;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* ap = As + (tm * 16 + frow) * pitch + fk;
    const bf16_t* bp = Bs + (tn * 16 + frow) * pitch + fk;
    for (int k0 = 0; k0 < Kp; k0 += 32) {
      const bf16x8 af = *(const bf16x8*)(ap + k0), bfr = *(const bf16x8*)(bp + k0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr, af, acc, 0, 0, 0);   // D^T: lane = (row m = lane&15, cols n = 4*(lane>>4)+r)
    }
    const int m = tm * 16 + (lane & 15), n0 = tn * 16 + (lane >> 4) * 4;
    if (m < g.M && n0 < g.N) {
      TO* o = out + (int64_t)m * ep.ldo + n0;
      const float4 v = make_float4(acc[0] * ep.alpha, acc[1] * ep.alpha, acc[2] * ep.alpha, acc[3] * ep.alpha);
      if (n0 + 3 < g.N && ep.vec_ok) st4<TO>(o, v);
      else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int i = 0; i < 4 && n0 + i < g.N; ++i) stf<TO>(o + i, vv[i]);
      }
    }
  }
}

template <typename TA, typename TO>
__global__ __launch_bounds__(256) void bgemm_mfma_kernel(GenericGemmArgs g, EpiParams ep, int Mp, int Np, int Kp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bgemm_mfma_body<TA, TO>(g, ep, Mp, Np, Kp, smem);
}

// Two products of the same (image, head) in one launch, run back to back by the same workgroup: the attention backward's pairs
// (dA = dO V^T, dV = A^T dO) and (dQ = dS K, dK = dS^T Q) share an operand -- dO, resp. the [b, h, n, n] fp32 dS -- which the second
// product then finds in L2 instead of fetching it from HBM again (same idea as attn_bwd_fused_kernel), and a launch is saved.
template <typename TA1, typename TO1, typename TA2, typename TO2>
__global__ __launch_bounds__(256) void bgemm_mfma_pair_kernel(GenericGemmArgs g1, EpiParams ep1, int Mp1, int Np1, int Kp1, GenericGemmArgs g2,
                                                              EpiParams ep2, int Mp2, int Np2, int Kp2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bgemm_mfma_body<TA1, TO1>(g1, ep1, Mp1, Np1, Kp1, smem);
  __syncthreads();                     // every wave is done reading the first product's LDS images
  bgemm_mfma_body<TA2, TO2>(g2, ep2, Mp2, Np2, Kp2, smem);
}

}  // namespace

bool bgemm_mfma_supported(const GenericGemmArgs& g, int ta, int tb, int to, int mode) {
  if (g.M > 128 || g.N > 128 || g.K > 128 || g.M < 1 || g.N < 1 || g.K < 1) return false;
  if (tb != 1) return false;                                           // B operand: bf16 q / k / v / dO head slices
  // A: bf16 head slices (q, k, v, dO) or bf16 score tensors kept by the one-kernel forwards (round 5) -> fp32 scores or bf16 rows; or fp32 scores -> bf16 rows
  if (!((ta == 1 && to == 0 && mode == EPI_STORE_F32) || (to == 1 && mode == EPI_STORE))) return false;
  return true;
}

void launch_bgemm_mfma_pair(const GenericGemmArgs& g1, const EpiParams& ep1, int ta1, int to1, const GenericGemmArgs& g2, const EpiParams& ep2, int ta2,
                            int to2, hipStream_t s) {
  const int Mp1 = (int)round_up(g1.M, 16), Np1 = (int)round_up(g1.N, 16), Kp1 = (int)round_up(g1.K, 32);
  const int Mp2 = (int)round_up(g2.M, 16), Np2 = (int)round_up(g2.N, 16), Kp2 = (int)round_up(g2.K, 32);
  const size_t smem = std::max((size_t)(Mp1 + Np1) * (Kp1 + PAD) * 2, (size_t)(Mp2 + Np2) * (Kp2 + PAD) * 2);
  dim3 grid((unsigned)(g1.nb * g1.nh)), block(256);
#define VITX_PAIR(T1, O1, T2, O2)                                                                                                  \
  {                                                                                                                                \
    auto kern = bgemm_mfma_pair_kernel<T1, O1, T2, O2>;                                                                            \
    vitx_set_max_smem((const void*)kern, 80 * 1024);                                                                               \
    hipLaunchKernelGGL(kern, grid, block, smem, s, g1, ep1, Mp1, Np1, Kp1, g2, ep2, Mp2, Np2, Kp2);                                \
  }
  if (ta1 && to1 && ta2 && to2) VITX_PAIR(bf16_t, bf16_t, bf16_t, bf16_t)      // (dV = A'^T dO, dK = dS^T q) / (dQ = dS k, dK = dS^T q) on bf16 score tensors
  else if (ta1 && !to1 && ta2 && to2) VITX_PAIR(bf16_t, float, bf16_t, bf16_t)   // (dA = dO v^T, dV = A'^T dO) with A' kept as bf16
  else if (ta1 && !ta2) VITX_PAIR(bf16_t, float, float, bf16_t)
  else if (!ta1 && !ta2) VITX_PAIR(float, bf16_t, float, bf16_t)
  else if (ta1 && ta2) VITX_PAIR(bf16_t, float, bf16_t, float)
  else VITX_PAIR(float, bf16_t, bf16_t, float)
#undef VITX_PAIR
}

void launch_bgemm_mfma(const GenericGemmArgs& g, const EpiParams& ep, int ta, int to, hipStream_t s) {
  const int Mp = (int)round_up(g.M, 16), Np = (int)round_up(g.N, 16), Kp = (int)round_up(g.K, 32);
  const size_t smem = (size_t)(Mp + Np) * (Kp + PAD) * 2;
  dim3 grid((unsigned)(g.nb * g.nh)), block(256);
  if (ta && to) {
    auto kern = bgemm_mfma_kernel<bf16_t, bf16_t>;
    vitx_set_max_smem((const void*)kern, 80 * 1024);
    hipLaunchKernelGGL(kern, grid, block, smem, s, g, ep, Mp, Np, Kp);
  } else if (ta) {
    auto kern = bgemm_mfma_kernel<bf16_t, float>;
    vitx_set_max_smem((const void*)kern, 80 * 1024);
    hipLaunchKernelGGL(kern, grid, block, smem, s, g, ep, Mp, Np, Kp);
  } else {
    auto kern = bgemm_mfma_kernel<float, bf16_t>;
    vitx_set_max_smem((const void*)kern, 80 * 1024);
    hipLaunchKernelGGL(kern, grid, block, smem, s, g, ep, Mp, Np, Kp);
  }
}
