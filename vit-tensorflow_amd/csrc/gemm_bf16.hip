// bf16 MFMA GEMM for the Dense layers of the ViT path (vit.py:39,42,59,63,143,156 and their VJPs):
//   C[M,N] = A[M,K] * B[N,K]^T   ("NT": both operands K-contiguous), fp32 accumulation.
// gfx950 design:
//   * v_mfma_f32_32x32x16_bf16, operands swapped (mfma(Bfrag, Afrag)) so that each lane ends up with
//     4 consecutive output columns of one row -> 8/16-byte epilogue accesses, fused epilogues.
//   * direct-to-LDS loads (global_load_lds_dwordx4, 1 KiB per wave instruction), double-buffered
//     BK = 64 stages, one barrier per K-tile; the next tile's DMA is in flight during the MFMAs.
//   * LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is applied to the
//     per-lane SOURCE address and to the ds_read address: chunk ^= (row >> 1) & 7  (16-B chunks of a
//     128-B row) -- conflict-free for the 16-lane groups ds_read_b128 is serviced in.
//   * XCD-aware tile order: the grid is walked so that consecutive logical tiles (sharing an A row
//     panel) run on the same XCD and hit its private L2.
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

constexpr int BK = 64;

template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_nt_kernel(Bf16GemmArgs g, EpiParams ep, int tiles_m, int tiles_n,
                                                                    int kt_per_split) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- XCD-aware tile mapping (bijective for any grid size)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
  const int z = blockIdx.y;
  const int nk_total = g.K / BK;
  const int kt0 = z * kt_per_split;
  const int nk = min(kt_per_split, nk_total - kt0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  // ---- per-lane DMA source offsets (elements); LDS destination is wave-uniform base + lane*16
  const bf16_t* Ag = g.A + (int64_t)tile_m * BM * g.lda + (int64_t)kt0 * BK;
  const bf16_t* Bg = g.B + (int64_t)tile_n * BN * g.ldb + (int64_t)kt0 * BK;
  int64_t offA[A_INSTR], offB[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    offA[i] = (int64_t)row * g.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    offB[i] = (int64_t)row * g.ldb + c * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ag + offA[i] + (int64_t)kt * BK),
                                       (lds_void_t*)(base + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Bg + offB[i] + (int64_t)kt * BK),
                                       (lds_void_t*)(base + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read addressing: row = tile row of this lane (lane&31), chunk = (ks*2 + (lane>>5)) ^ swz(row)
  const int sw = ((lane & 31) >> 1) & 7;
  const int a_row_byte = (wm * WTM + (lane & 31)) * 128;
  const int b_row_byte = A_BYTES + (wn * WTN + (lane & 31)) * 128;
  const int khalf = lane >> 5;

  if (nk > 0) stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* base = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int cb = ((ks * 2 + khalf) ^ sw) << 4;
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8*)(base + a_row_byte + i * 32 * 128 + cb);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = *(const bf16x8*)(base + b_row_byte + j * 32 * 128 + cb);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds row m = lane&31 and columns 8q + 4*(lane>>5) + {0..3} of each 32x32 tile
  const int64_t out_off = (int64_t)z * ep.partial_stride;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = tile_m * BM + wm * WTM + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col0 = tile_n * BN + wn * WTN + j * 32 + 4 * khalf;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        epilogue_apply4<MODE, bf16_t>(ep, row, col0 + 8 * q,
                                      make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]),
                                      out_off);
    }
  }
}

template <int BM, int BN, int WM, int WN, int MODE>
void launch_variant(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  constexpr int SMEM = 2 * (BM + BN) * BK * 2;
  auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, MODE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  const int tiles_m = (int)ceil_div(g.M, BM), tiles_n = (int)ceil_div(g.N, BN);
  const int nk = g.K / BK;
  const int split = g.split_k > 1 ? g.split_k : 1;
  const int per = (int)ceil_div(nk, split);
  const int zs = (int)ceil_div(nk, per);  // no empty slices
  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)zs), block(WM * WN * 64);
  hipLaunchKernelGGL(kern, grid, block, SMEM, s, g, ep, tiles_m, tiles_n, per);
}

template <int MODE>
void launch_mode(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  int k = g.kernel;
  if (k == 0) k = (g.N % 256 == 0 || g.N > 512) ? 2 : 1;
  if (k == 1) launch_variant<128, 128, 2, 2, MODE>(g, ep, s);
  else if (k == 3) launch_variant<256, 128, 4, 2, MODE>(g, ep, s);
  else launch_variant<256, 256, 2, 4, MODE>(g, ep, s);
}

}  // namespace

int gemm_bf16_tile_m(int kernel, int M, int N) {
  if (kernel == 0) kernel = (N % 256 == 0 || N > 512) ? 2 : 1;
  return kernel == 1 ? 128 : 256;
}
int gemm_bf16_tile_n(int kernel, int M, int N) {
  if (kernel == 0) kernel = (N % 256 == 0 || N > 512) ? 2 : 1;
  return kernel == 2 ? 256 : 128;
}

// Number of K slices a split-K launch actually produces (matches launch_variant)
int gemm_bf16_num_slices(int K, int split_k) {
  const int nk = K / BK;
  const int split = split_k > 1 ? split_k : 1;
  const int per = (int)ceil_div(nk, split);
  return (int)ceil_div(nk, per);
}

void launch_gemm_bf16(const Bf16GemmArgs& g, const EpiParams& ep, int mode, hipStream_t s) {
  switch (mode) {
    case EPI_STORE: launch_mode<EPI_STORE>(g, ep, s); break;
    case EPI_STORE_F32: launch_mode<EPI_STORE_F32>(g, ep, s); break;
    case EPI_BIAS_GELU: launch_mode<EPI_BIAS_GELU>(g, ep, s); break;
    case EPI_BIAS_RESID: launch_mode<EPI_BIAS_RESID>(g, ep, s); break;
    case EPI_PATCH: launch_mode<EPI_PATCH>(g, ep, s); break;
    case EPI_GELU_BWD: launch_mode<EPI_GELU_BWD>(g, ep, s); break;
    case EPI_PARTIAL: launch_mode<EPI_PARTIAL>(g, ep, s); break;
    default: break;
  }
}
