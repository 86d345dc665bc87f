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
#include <type_traits>

#include "kernels.h"

static int g_allow_320 = 1;
int gemm_bf16_pick(int M, int N);

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

constexpr int BK = 64;

template <int BM, int BN, int WM, int WN, int MODE, bool LDS_EPI, int SCHED>
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

  // Optional phase stagger: every workgroup runs [K loop (MFMA-bound)] -> [epilogue (HBM-bound)]; launched together
  // they stay in lockstep and the two resources alternate idling.  Delaying the first wave of workgroups by a quarter
  // period per CU slot spreads the phases; later workgroups inherit the offset of the CU they land on.
  if (g.stagger > 0 && bid < 256 && blockIdx.y == 0) {
    const int ph = (bid >> 3) & 3;
    for (int i = 0; i < ph * g.stagger; ++i) __builtin_amdgcn_s_sleep(32);
  }

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

  if constexpr (SCHED == 0) {
    // ---- lockstep schedule: one barrier per K-tile, next tile's DMA in flight during the MFMAs
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
  } else {
    // ---- ping-pong schedule (8 waves = 2 per SIMD).  A K-tile is processed as two halves (k 0..31, k 32..63), each half as a
    // LOAD phase (ds_read the fragments, issue DMA) and an MFMA phase (16 MFMAs), separated by workgroup barriers.  The second
    // wave of every SIMD (waves NW/2..NW-1, "group B") runs the same sequence ONE PHASE LATE, so at any time one wave of a SIMD is
    // in its MFMA phase while its partner reads LDS / issues DMA / waits: the matrix pipe is no longer idle around the barriers.
    //   epoch:      4t        4t+1      4t+2      4t+3      4t+4
    //   group A:  L(t,H0)   M(t,H0)   L(t,H1)   M(t,H1)   L(t+1,H0)      A issues its DMA share of tile t+1 at L(t,H0)
    //   group B:  M(t-1,H1) L(t,H0)   M(t,H0)   L(t,H1)   M(t,H1)        B issues its DMA share of tile t+1 at M(t-1,H1)
    // Buffer (t+1)&1 was last read in epoch 4t-1 (B's L(t-1,H1)), so both DMA issues (epoch 4t) are WAR-safe; tile t+1 is first
    // read in epoch 4t+4, and every wave drains its own DMA (vmcnt(0)) before the barrier that ends epoch 4t+3.
    static_assert(NW == 8, "ping-pong needs two waves per SIMD");
    const bool grpB = wave >= NW / 2;
    bf16x8 af[2][MT], bfr[2][NT];
    auto load_half = [&](const char* base, int h) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int cb = (((2 * h + kk) * 2 + khalf) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) af[kk][i] = *(const bf16x8*)(base + a_row_byte + i * 32 * 128 + cb);
#pragma unroll
        for (int j = 0; j < NT; ++j) bfr[kk][j] = *(const bf16x8*)(base + b_row_byte + j * 32 * 128 + cb);
      }
    };
    auto mma_half = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
#define VITX_BAR()                                  \
  do {                                              \
    asm volatile("" ::: "memory");                  \
    __builtin_amdgcn_s_barrier();                   \
    asm volatile("" ::: "memory");                  \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)
    if (nk > 0) stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VITX_BAR();                                           // tile 0 visible to every wave
    if (grpB) {
      if (nk > 1) stage(1, 1);                            // B's share of tile 1 (its "M(-1,H1)" slot)
      VITX_BAR();                                         // B starts one epoch late
    }
    for (int t = 0; t < nk; ++t) {
      const char* base = smem + (t & 1) * STAGE;
      // ---- L(t,H0)
      if (!grpB && t + 1 < nk) stage((t + 1) & 1, t + 1);
      load_half(base, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      VITX_BAR();
      // ---- M(t,H0)
      mma_half();
      VITX_BAR();
      // ---- L(t,H1)
      load_half(base, 1);
      if (grpB) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      VITX_BAR();
      // ---- M(t,H1)
      if (grpB && t + 2 < nk) stage(t & 1, t + 2);
      mma_half();
      if (!grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      VITX_BAR();
    }
    if (!grpB) VITX_BAR();                                // balance B's extra barrier
#undef VITX_BAR
  }

  // ---- epilogue: lane holds row m = lane&31 and columns 8q + 4*(lane>>5) + {0..3} of each 32x32 tile
  const int64_t out_off = (int64_t)z * ep.partial_stride;
  // interior tiles (all but the ragged edge) take the branch-free epilogue: every wave-uniform test is made once, here
  const bool interior = epilogue_fast_ok(ep, MODE) && (tile_m + 1) * BM <= ep.M && (tile_n + 1) * BN <= ep.N;
  const bool has_bias = ep.bias != nullptr, has_scale = ep.scale != nullptr;
  if constexpr (LDS_EPI && (MT % 2 == 0)) {
    // Stage 64 output rows at a time through LDS (fp32, padded rows) and run the fused epilogue on ROW-CONTIGUOUS data:
    // every wave instruction then reads/writes whole 512-B / 1-KiB row segments (full cache lines) instead of
    // 32 scattered 16/32-B pieces.
    constexpr int SROW = BN + 4;                     // floats; +16 B keeps the 8-lane ds_write_b128 groups conflict-free
    constexpr int LPRW = BN / 4, RPIW = 64 / LPRW;   // lanes per staged row, rows per wave instruction
    constexpr int RPW = 64 / (NW * RPIW);            // rows handled per wave per round
    float* st = (float*)smem;
    const int col_l = (lane % LPRW) * 4;
    const int gcol = tile_n * BN + col_l;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (interior && has_bias) b4 = *(const float4*)(ep.bias + gcol);
    if (interior && has_scale) s4 = *(const float4*)(ep.scale + gcol);
#pragma unroll
    for (int R = 0; R < BM / 64; ++R) {
      const int wm_r = (R * 64) / WTM, i0 = ((R * 64) % WTM) / 32;
      __syncthreads();
      if (wm == wm_r) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *(float4*)(st + (ii * 32 + (lane & 31)) * SROW + wn * WTN + j * 32 + 8 * q + 4 * khalf) =
                  make_float4(acc[i0 + ii][j][4 * q], acc[i0 + ii][j][4 * q + 1], acc[i0 + ii][j][4 * q + 2], acc[i0 + ii][j][4 * q + 3]);
      }
      __syncthreads();
      float4 v[RPW];
#pragma unroll
      for (int k = 0; k < RPW; ++k) v[k] = *(const float4*)(st + ((k * NW + wave) * RPIW + lane / LPRW) * SROW + col_l);
      const int grow0 = tile_m * BM + R * 64 + lane / LPRW;
      if (interior) {
        float4 x[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) x[k] = epilogue_fast_load<MODE, bf16_t>(ep, grow0 + (k * NW + wave) * RPIW, gcol);
        if (has_bias) {
#pragma unroll
          for (int k = 0; k < RPW; ++k) epilogue_fast4<MODE, bf16_t, true, false>(ep, grow0 + (k * NW + wave) * RPIW, gcol, v[k], b4, s4, x[k], out_off);
        } else {
#pragma unroll
          for (int k = 0; k < RPW; ++k) epilogue_fast4<MODE, bf16_t, false, false>(ep, grow0 + (k * NW + wave) * RPIW, gcol, v[k], b4, s4, x[k], out_off);
        }
      } else {
#pragma unroll
        for (int k = 0; k < RPW; ++k) epilogue_apply4<MODE, bf16_t>(ep, grow0 + (k * NW + wave) * RPIW, gcol, v[k], out_off);
      }
    }
  } else {
    const int row0 = tile_m * BM + wm * WTM + (lane & 31);
    const int col00 = tile_n * BN + wn * WTN + 4 * khalf;
    if (interior) {
      auto run = [&](auto hb, auto hs) {
        constexpr bool HB = decltype(hb)::value, HS = decltype(hs)::value;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {       // two column groups at a time: 2*MT global reads in flight before the stores
            float4 b4[2], s4[2], x[2][MT];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
              const int col = col00 + j * 32 + 8 * (2 * qp + qq);
              b4[qq] = HB ? *(const float4*)(ep.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
              s4[qq] = HS ? *(const float4*)(ep.scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
              for (int i = 0; i < MT; ++i) x[qq][i] = epilogue_fast_load<MODE, bf16_t>(ep, row0 + i * 32, col);
            }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
              const int q = 2 * qp + qq;
              const int col = col00 + j * 32 + 8 * q;
#pragma unroll
              for (int i = 0; i < MT; ++i)
                epilogue_fast4<MODE, bf16_t, HB, HS>(ep, row0 + i * 32, col,
                                                     make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]),
                                                     b4[qq], s4[qq], x[qq][i], out_off);
            }
          }
      };
      if (has_bias && has_scale) run(std::true_type{}, std::true_type{});
      else if (has_bias) run(std::true_type{}, std::false_type{});
      else run(std::false_type{}, std::false_type{});
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            epilogue_apply4<MODE, bf16_t>(ep, row0 + i * 32, col00 + j * 32 + 8 * q,
                                          make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]),
                                          out_off);
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int MODE, bool LDS_EPI, int SCHED = 0>
void launch_variant(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  constexpr int SMEM = 2 * (BM + BN) * BK * 2;
  static_assert(64 * (BN + 4) * 4 <= SMEM, "epilogue staging must fit in the pipeline buffers");
  auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, MODE, LDS_EPI, SCHED>;
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


// ------------------------------------------------------------------------------------------------
// "TN" GEMM for the weight gradients:  C[i][j] = sum_m A[m][i] * B[m][j]   (dW = X^T dY, reduction over token rows)
// Both operands are read in their natural row-major layout ([token][feature], feature contiguous); the
// transposition the MFMA needs (8 consecutive reduction indices per lane) is done by the LDS hardware
// transpose read ds_read_b64_tr_b16 (gfx950): within a 16-lane group, lanes 4j..4j+3 address 16 consecutive
// features of token row j, and lane q receives feature q of rows 0..3.  LDS tile = [64 tokens][BM features],
// filled by global_load_lds (lane-linear image), 32-B granules XOR-swizzled by 2*(token&3) on the DMA source and
// on the read address so that the 8 (row, granule) segments a half-wave touches fall in 8 distinct bank groups.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

template <int BT, int WM, int WN, int MODE>   // BT = tile extent in both feature dimensions
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_tn_kernel(Bf16GemmArgs g, EpiParams ep, int tiles_m, int tiles_n,
                                                                    int kt_per_split) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BT / WM, WTN = BT / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int ROWB = BT * 2;                 // bytes per token row of a tile
  constexpr int LPR = ROWB / 16;               // lanes (16-B chunks) per row
  constexpr int RPI = 64 / LPR;                // token rows per wave DMA instruction
  constexpr int OP_BYTES = BK * ROWB, STAGE = 2 * OP_BYTES;
  constexpr int INSTR = BK / RPI / NW;         // DMA instructions per wave per operand per stage
  static_assert(BK % (RPI * NW) == 0, "token rows must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

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

  // DMA: instruction i of this wave covers token rows (i*NW + wave)*RPI .. +RPI-1; lane -> (row, physical chunk)
  const bf16_t* Ag = g.A + (int64_t)kt0 * BK * g.lda + (int64_t)tile_m * BT;
  const bf16_t* Bg = g.B + (int64_t)kt0 * BK * g.ldb + (int64_t)tile_n * BT;
  int64_t offA[INSTR], offB[INSTR];
#pragma unroll
  for (int i = 0; i < INSTR; ++i) {
    const int row = (i * NW + wave) * RPI + lane / LPR;
    const int pc = lane % LPR;
    const int c = ((((pc >> 1) ^ (2 * (row & 3))) << 1) | (pc & 1));   // logical 16-B chunk stored at physical chunk pc
    offA[i] = (int64_t)row * g.lda + c * 8;
    offB[i] = (int64_t)row * g.ldb + c * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < INSTR; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ag + offA[i] + (int64_t)kt * BK * g.lda),
                                       (lds_void_t*)(base + (i * NW + wave) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Bg + offB[i] + (int64_t)kt * BK * g.ldb),
                                       (lds_void_t*)(base + OP_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transpose-read addressing: lane = 16*G + q; G&1 -> 16-feature sub-block, G>>1 -> k-half; q>>2 -> token row, q&3 -> 8-B piece
  const int q = lane & 15, G = lane >> 4, khalf = lane >> 5;
  const int trow = q >> 2;                                  // token row within the group of 4
  // byte offset inside a token row of this lane's 8-B piece, before swizzle, for feature block fb (32 features = 64 B)
  const int piece = (G & 1) * 32 + (q & 3) * 8;             // bytes within the 64-B span of a 32-feature block
  auto row_addr = [&](int m, int feat_byte) {               // swizzle the 32-B granule index by 2*(m&3)
    const int gran = (feat_byte >> 5) ^ (2 * (m & 3));
    return m * ROWB + (gran << 5) + (feat_byte & 31);
  };

  if (nk > 0) stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* base = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int m0 = 16 * ks + 8 * khalf + trow, m1 = m0 + 4;
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int fb = (wm * WTM + i * 32) * 2 + piece;
        af[i] = tr_frag(base + row_addr(m0, fb), base + row_addr(m1, fb));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int fb = (wn * WTN + j * 32) * 2 + piece;
        bfr[j] = tr_frag(base + OP_BYTES + row_addr(m0, fb), base + OP_BYTES + row_addr(m1, fb));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  const int64_t out_off = (int64_t)z * ep.partial_stride;
  const bool interior = epilogue_fast_ok(ep, MODE) && (tile_m + 1) * BT <= ep.M && (tile_n + 1) * BT <= ep.N;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = tile_m * BT + wm * WTM + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col0 = tile_n * BT + wn * WTN + j * 32 + 4 * khalf;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float4 v = make_float4(acc[i][j][4 * qq], acc[i][j][4 * qq + 1], acc[i][j][4 * qq + 2], acc[i][j][4 * qq + 3]);
        if (interior) epilogue_fast4<MODE, bf16_t, false, false>(ep, row, col0 + 8 * qq, v, z4, z4, z4, out_off);
        else epilogue_apply4<MODE, bf16_t>(ep, row, col0 + 8 * qq, v, out_off);
      }
    }
  }
}

template <int BT, int WM, int WN>
void launch_tn_variant(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  constexpr int SMEM = 2 * 2 * BK * BT * 2;
  auto kern = gemm_bf16_tn_kernel<BT, WM, WN, EPI_PARTIAL>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  const int tiles_m = (int)ceil_div(g.M, BT), tiles_n = (int)ceil_div(g.N, BT);
  const int nk = g.K / BK;
  const int split = g.split_k > 1 ? g.split_k : 1;
  const int per = (int)ceil_div(nk, split);
  const int zs = (int)ceil_div(nk, per);
  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)zs), block(WM * WN * 64);
  hipLaunchKernelGGL(kern, grid, block, SMEM, s, g, ep, tiles_m, tiles_n, per);
}

template <int MODE>
void launch_mode(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  int k = g.kernel & 15;
  if (k == 0) k = gemm_bf16_pick(g.M, g.N);
  // epilogue form: bit 8 forces the per-lane direct form, bit 9 forces the LDS-staged form; by default bf16-output epilogues
  // (8-B per-lane pieces) are staged through LDS into whole-row stores, fp32-output ones (16-B pieces) go out directly (measured).
  const bool bf16_out = (MODE == EPI_STORE || MODE == EPI_BIAS_GELU || MODE == EPI_GELU_BWD);
  const bool direct = (g.kernel & 256) ? true : ((g.kernel & 512) ? false : !bf16_out);
  if (direct) {
    if (k == 1) launch_variant<128, 128, 2, 2, MODE, false>(g, ep, s);
    else if (k == 3) launch_variant<256, 128, 4, 2, MODE, false>(g, ep, s);
    else if (k == 4) launch_variant<256, 256, 2, 4, MODE, false, 1>(g, ep, s);
    else if (k == 5) launch_variant<320, 256, 2, 4, MODE, false>(g, ep, s);
    else launch_variant<256, 256, 2, 4, MODE, false>(g, ep, s);
  } else {
    if (k == 1) launch_variant<128, 128, 2, 2, MODE, true>(g, ep, s);
    else if (k == 3) launch_variant<256, 128, 4, 2, MODE, true>(g, ep, s);
    else if (k == 4) launch_variant<256, 256, 2, 4, MODE, true, 1>(g, ep, s);
    else if (k == 5) launch_variant<320, 256, 2, 4, MODE, false>(g, ep, s);   // MT = 5: direct epilogue only
    else launch_variant<256, 256, 2, 4, MODE, true>(g, ep, s);
  }
}

}  // namespace

// automatic tile choice: 128x128 for small N; otherwise 256x256, or 320x256 when that fills the 256 CUs' rounds better
// (M = 50432, N = 768: 591 tiles = 2.31 rounds (77 %) vs 474 tiles = 1.85 rounds (93 %)).
int gemm_bf16_pick(int M, int N) {
  if (!(N % 256 == 0 || N > 512)) return 1;
  const int64_t tn = ceil_div(N, 256);
  const int64_t t256 = ceil_div(M, 256) * tn, t320 = ceil_div(M, 320) * tn;
  const double e256 = (double)M * N / ((double)ceil_div(t256, 256) * 256 * 256 * 256);
  const double e320 = (double)M * N / ((double)ceil_div(t320, 256) * 256 * 320 * 256);
  return (g_allow_320 && e320 > e256 * 1.08) ? 5 : 2;
}
int gemm_bf16_tile_m(int kernel, int M, int N) {
  kernel &= 15;
  if (kernel == 0) kernel = gemm_bf16_pick(M, N);
  return kernel == 1 ? 128 : (kernel == 5 ? 320 : 256);
}
int gemm_bf16_tile_n(int kernel, int M, int N) {
  kernel &= 15;
  if (kernel == 0) kernel = gemm_bf16_pick(M, N);
  return (kernel == 1 || kernel == 3) ? 128 : 256;
}
void gemm_bf16_allow_320(int on) { g_allow_320 = on; }

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

// C[M=in][N=out] (split-K partials) = A[K=tokens][in]^T * B[K=tokens][out]; kernel: 1 = 128x128 tile, else 256x256
int gemm_bf16_tn_tile(int kernel, int M, int N) { kernel &= 15; return (kernel == 1 || (M <= 128 && N <= 128)) ? 128 : 256; }
void launch_gemm_bf16_tn(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  if (gemm_bf16_tn_tile(g.kernel, g.M, g.N) == 128) launch_tn_variant<128, 2, 2>(g, ep, s);
  else launch_tn_variant<256, 2, 4>(g, ep, s);
}
