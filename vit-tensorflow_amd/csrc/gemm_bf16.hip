// bf16 MFMA GEMMs for the Dense layers of the ViT path (vit.py:39,42,59,63,143,156 and their VJPs), fp32 accumulation.
//   NT family:  C[M,N] = A[M,K] * B[N,K]^T   (both operands K-contiguous)     forward + input gradients, fused epilogues
//   TN kernel:  C[M,N] = A[K,M]^T * B[K,N]   (both operands token-major)      weight gradients, split-K partials
// gfx950 design, shared by every kernel in this file:
//   * v_mfma_f32_32x32x16_bf16, operands swapped (mfma(Bfrag, Afrag)) so that each lane ends up with
//     4 consecutive output columns of one row -> 8/16-byte epilogue accesses, fused epilogues.
//   * direct-to-LDS loads (1 KiB per wave instruction), double-buffered BK = 64 stages, one barrier per K-tile.
//   * the LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is applied to the
//     per-lane SOURCE address and to the ds_read address: chunk ^= (row >> 1) & 7  (16-B chunks of a
//     128-B row) -- conflict-free for the 16-lane groups ds_read_b128 is serviced in.
//   * XCD-aware work order: the workgroup list is walked so that consecutive logical tiles (sharing an A row
//     panel / a K-slice) run on the same XCD and hit its private L2.
// Kernels:
//   gemm_bf16_nt_kernel        one tile per workgroup (SCHED 0) or persistent with cross-tile prefetch (SCHED 2), lockstep K loop
//   gemm_bf16_nt_pipe_kernel   persistent, register double-buffered fragments, DMA pieces spread over the K-tile, buffer-addressed
//   gemm_bf16_tn_kernel        transpose-read (ds_read_b64_tr_b16) fragments, same pipelined loop, K-slice-major 1-D grid
//   launch_gemm_bf16           variant dispatch; kernel = 0 measures the candidates once per (epilogue, M, N, K) and caches the winner
#include <array>
#include <map>
#include <mutex>

#include "gemm_bf16_common.h"

static int g_allow_320 = 1;
static int g_shared_gpu = 0;
int gemm_bf16_pick(int M, int N);

namespace {

template <int BM, int BN, int WM, int WN, int MODE, bool LDS_EPI, int SCHED>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_nt_kernel(Bf16GemmArgs g, EpiParams ep, int tiles_m, int tiles_n,
                                                                    int kt_per_split) {
  // SCHED 0: one tile per workgroup, lockstep 2-stage K loop.
  // SCHED 2: PERSISTENT workgroups (grid = #CUs) walking the tile list; the first K-tile of the NEXT output tile is prefetched
  //          during the last K-tile of the current one, so the DMA latency and the epilogue's memory phase overlap.
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
  const int logical0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int total_tiles = tiles_m * tiles_n;
  const int z = blockIdx.y;
  const int nk_total = g.K / BK;
  const int kt0 = z * kt_per_split;
  const int nk = min(kt_per_split, nk_total - kt0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  if (g.stagger > 0 && bid < 256 && blockIdx.y == 0) {   // optional phase stagger of the first wave of workgroups (experiment)
    const int ph = (bid >> 3) & 3;
    for (int i = 0; i < ph * g.stagger; ++i) __builtin_amdgcn_s_sleep(32);
  }

  // ---- per-lane DMA source offsets (elements); LDS destination is wave-uniform base + lane*16
  int offA[A_INSTR], offB[B_INSTR];   // 32-bit: a tile spans < 2^31 elements of its operand
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    offA[i] = row * (int)g.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    offB[i] = row * (int)g.ldb + c * 8;
  }
  auto stage_ptr = [&](int buf, const bf16_t* Ap, const bf16_t* Bp, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ap + (offA[i] + kt * BK)),
                                       (lds_void_t*)(base + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(Bp + (offB[i] + kt * BK)),
                                       (lds_void_t*)(base + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
  };

  // fragment read addressing: row = tile row of this lane (lane&31), chunk = (ks*2 + (lane>>5)) ^ swz(row)
  const int sw = ((lane & 31) >> 1) & 7;
  const int a_row_byte = (wm * WTM + (lane & 31)) * 128;
  const int b_row_byte = A_BYTES + (wn * WTN + (lane & 31)) * 128;
  const int khalf = lane >> 5;
  const bool has_bias = ep.bias != nullptr, has_scale = ep.scale != nullptr;
  const int64_t out_off = (int64_t)z * ep.partial_stride;

  int it = 0;                                   // running K-tile counter: LDS buffer = it & 1 (continues across output tiles)
  int logical = logical0;
  if (logical >= total_tiles) return;
  const int gm = (tiles_n >= 8 ? 4 : 1) * (g.reverse_m ? -1 : 1);
  int tile_m, tile_n;
  decode_tile(logical, tiles_m, tiles_n, gm, tile_m, tile_n);
  const bf16_t* Ag = g.A + (int64_t)tile_m * BM * g.lda + (int64_t)kt0 * BK;
  const bf16_t* Bg = g.B + (int64_t)tile_n * BN * g.ldb + (int64_t)kt0 * BK;
  if (SCHED == 2 && nk > 0) stage_ptr(0, Ag, Bg, 0);

  for (;;) {
    auto stage = [&](int buf, int kt) { stage_ptr(buf, Ag, Bg, kt); };
    // next output tile of this (persistent) workgroup
    const int next_logical = logical + nwg;
    const bool has_next = SCHED == 2 && next_logical < total_tiles;
    int ntm = 0, ntn = 0;
    if (has_next) decode_tile(next_logical, tiles_m, tiles_n, gm, ntm, ntn);
    const bf16_t* Agn = g.A + (int64_t)ntm * BM * g.lda + (int64_t)kt0 * BK;
    const bf16_t* Bgn = g.B + (int64_t)ntn * BN * g.ldb + (int64_t)kt0 * BK;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    {
      // ---- lockstep schedule: one barrier per K-tile, next tile's DMA in flight during the MFMAs
      if (SCHED != 2 && nk > 0) stage(0, 0);
      for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((it + 1) & 1, kt + 1);
        else if (has_next) stage_ptr((it + 1) & 1, Agn, Bgn, 0);
        const char* base = smem + (it & 1) * STAGE;
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
        ++it;
      }
    }

    // ---- epilogue: lane holds row m = lane&31 and columns 8q + 4*(lane>>5) + {0..3} of each 32x32 tile.
    // Interior tiles (all but the ragged edge) take the branch-free form: every wave-uniform test is made once, here.
    const bool interior = epilogue_fast_ok(ep, MODE) && (tile_m + 1) * BM <= ep.M && (tile_n + 1) * BN <= ep.N;
    // rows per LDS staging round: 64, or 32 when only one pipeline buffer is free (persistent schedule) or MT is odd
    constexpr int RR = ((SCHED == 2 && 64 * (BN + 4) * 4 > STAGE) || (MT % 2 != 0)) ? 32 : 64;
    if constexpr (LDS_EPI) {
      // Stage RR output rows at a time through LDS (fp32, padded rows) and run the fused epilogue on ROW-CONTIGUOUS data:
      // every wave instruction then reads/writes whole 512-B / 1-KiB row segments (full cache lines) instead of
      // 32 scattered 16/32-B pieces.
      constexpr int SROW = BN + 4;                     // floats; +16 B keeps the 8-lane ds_write_b128 groups conflict-free
      constexpr int LPRW = BN / 4, RPIW = 64 / LPRW;   // lanes per staged row, rows per wave instruction
      constexpr int RPW = RR / (NW * RPIW);            // rows handled per wave per round
      static_assert(RPW >= 1 && RR * SROW * 4 <= (SCHED == 2 ? 1 : 2) * STAGE, "staging round must fit the free pipeline buffer(s)");
      // SCHED 2: buffer (it & 1) holds the prefetched next tile; the one just consumed is free
      float* st = (float*)(smem + (SCHED == 2 ? ((it + 1) & 1) * STAGE : 0));
      const int col_l = (lane % LPRW) * 4;
      const int gcol = tile_n * BN + col_l;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (interior && has_bias) b4 = *(const float4*)(ep.bias + gcol);
      if (interior && has_scale) s4 = *(const float4*)(ep.scale + gcol);
      float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);   // EPI_GELU_BWD: column sums of what this lane stores (fused bias gradient)
      auto cs_add = [&](float4 r) { if (MODE == EPI_GELU_BWD) { cs.x += r.x; cs.y += r.y; cs.z += r.z; cs.w += r.w; } };
#pragma clang loop unroll(full)
      for (int R = 0; R < BM / RR; ++R) {
        const int wm_r = (R * RR) / WTM, i0 = ((R * RR) % WTM) / 32;
        __syncthreads();
        if (wm == wm_r) {
#pragma unroll
          for (int ii = 0; ii < RR / 32; ++ii)
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
        const int grow0 = tile_m * BM + R * RR + lane / LPRW;
        if (interior) {
          float4 x[RPW];
#pragma unroll
          for (int k = 0; k < RPW; ++k) x[k] = epilogue_fast_load<MODE, bf16_t>(ep, grow0 + (k * NW + wave) * RPIW, gcol);
          if (MODE == EPI_BIAS_RESID && has_scale) {   // LayerScale (cait.py:47-48): f(x) kept in out2, the branch scaled per column
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
              if (has_bias) epilogue_fast4<MODE, bf16_t, true, true>(ep, grow0 + (k * NW + wave) * RPIW, gcol, v[k], b4, s4, x[k], out_off);
              else epilogue_fast4<MODE, bf16_t, false, true>(ep, grow0 + (k * NW + wave) * RPIW, gcol, v[k], b4, s4, x[k], out_off);
            }
          } else if (has_bias) {
#pragma unroll
            for (int k = 0; k < RPW; ++k) cs_add(epilogue_fast4<MODE, bf16_t, true, false>(ep, grow0 + (k * NW + wave) * RPIW, gcol, v[k], b4, s4, x[k], out_off));
          } else {
#pragma unroll
            for (int k = 0; k < RPW; ++k) cs_add(epilogue_fast4<MODE, bf16_t, false, false>(ep, grow0 + (k * NW + wave) * RPIW, gcol, v[k], b4, s4, x[k], out_off));
          }
        } else {
#pragma unroll
          for (int k = 0; k < RPW; ++k) cs_add(epilogue_apply4<MODE, bf16_t>(ep, grow0 + (k * NW + wave) * RPIW, gcol, v[k], out_off));
        }
      }
      if (MODE == EPI_GELU_BWD && ep.colsum != nullptr) {
        // per-tile column sums: NW*RPIW partial rows through the staging buffer, then one row of [BN] to colsum[tile_m][...]
        __syncthreads();
        *(float4*)(st + (wave * RPIW + lane / LPRW) * BN + col_l) = cs;
        __syncthreads();
        if (tid < BN) {
          float a = 0.f;
#pragma unroll
          for (int w = 0; w < NW * RPIW; ++w) a += st[w * BN + tid];
          const int c = tile_n * BN + tid;
          if (c < ep.N) ep.colsum[(int64_t)tile_m * ep.ldcs + c] = a;
        }
      }
      if (SCHED == 2) __syncthreads();   // the staging buffer becomes the next DMA target at the top of the next K loop
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

    if (!has_next) break;
    logical = next_logical;
    tile_m = ntm; tile_n = ntn;
    Ag = Agn; Bg = Bgn;
  }
}

template <int BM, int BN, int WM, int WN, int MODE, bool LDS_EPI, int SCHED = 0>
void launch_variant(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  constexpr int SMEM = 2 * (BM + BN) * BK * 2;
  static_assert(64 * (BN + 4) * 4 <= SMEM, "epilogue staging must fit in the pipeline buffers");
  auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, MODE, LDS_EPI, SCHED>;
  vitx_set_max_smem((const void*)kern, SMEM);
  const int tiles_m = (int)ceil_div(g.M, BM), tiles_n = (int)ceil_div(g.N, BN);
  const int nk = g.K / BK;
  const int split = g.split_k > 1 ? g.split_k : 1;
  const int per = (int)ceil_div(nk, split);
  const int zs = (int)ceil_div(nk, per);  // no empty slices
  const unsigned gx = (SCHED == 2 && zs == 1) ? (unsigned)std::min(tiles_m * tiles_n, 256) : (unsigned)(tiles_m * tiles_n);
  dim3 grid(gx, (unsigned)zs), block(WM * WN * 64);
  hipLaunchKernelGGL(kern, grid, block, SMEM, s, g, ep, tiles_m, tiles_n, per);
}

template <int MODE>
void launch_mode(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  int k = g.kernel & 15;
  if (k == 0) k = gemm_bf16_pick(g.M, g.N);
  // epilogue form: bit 8 forces the per-lane direct form, bit 9 forces the LDS-staged form; by default bf16-output epilogues
  // (8-B per-lane pieces) are staged through LDS into whole-row stores, fp32-output ones (16-B pieces) go out directly (measured).
  const bool bf16_out = (MODE == EPI_STORE || MODE == EPI_BIAS_GELU || MODE == EPI_GELU_BWD);
  // (8-B per-lane pieces) are staged; with 32-row staging rounds (variants 5-8) staging wins for fp32 outputs too.
  const bool direct = (g.kernel & 256) ? true : ((g.kernel & 512) ? false : (!bf16_out && k < 5));
  if (k == 9 || k == 10 || k == 11 || k == 13) { launch_gemm_bf16_pipe(k, MODE, g, ep, s); return; }   // gemm_bf16_pipe.hip
  if (direct) {
    if (k == 1) launch_variant<128, 128, 2, 2, MODE, false>(g, ep, s);
    else if (k == 3) launch_variant<256, 128, 4, 2, MODE, false>(g, ep, s);
    else if (k == 5) launch_variant<320, 256, 2, 4, MODE, false>(g, ep, s);
    else if (k == 6) launch_variant<256, 256, 2, 4, MODE, false, 2>(g, ep, s);
    else if (k == 7) launch_variant<320, 256, 2, 4, MODE, false, 2>(g, ep, s);
    else launch_variant<256, 256, 2, 4, MODE, false>(g, ep, s);
  } else {
    if (k == 1) launch_variant<128, 128, 2, 2, MODE, true>(g, ep, s);
    else if (k == 3) launch_variant<256, 128, 4, 2, MODE, true>(g, ep, s);
    else if (k == 5) launch_variant<320, 256, 2, 4, MODE, true>(g, ep, s);    // MT = 5: 32-row staging rounds
    else if (k == 6) launch_variant<256, 256, 2, 4, MODE, true, 2>(g, ep, s);
    else if (k == 7) launch_variant<320, 256, 2, 4, MODE, true, 2>(g, ep, s);
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
  if (g_shared_gpu) return (g_allow_320 && e320 > e256 * 1.08) ? 11 : 13;   // (one tile per workgroup beside collectives: launch_pipe)
  return (g_allow_320 && e320 > e256 * 1.08) ? 11 : 13;                    // pipelined persistent kernel (gemm_bf16_pipe.hip), 320 / 256-row tiles
}
int gemm_bf16_tile_m(int kernel, int M, int N) {
  kernel &= 15;
  if (kernel == 0) kernel = gemm_bf16_pick(M, N);
  return kernel == 1 ? 128 : ((kernel == 9 || kernel == 10) ? 192 : ((kernel == 5 || kernel == 7 || kernel == 11) ? 320 : 256));
}
int gemm_bf16_tile_n(int kernel, int M, int N) {
  kernel &= 15;
  if (kernel == 0) kernel = gemm_bf16_pick(M, N);
  return (kernel == 1 || kernel == 3 || kernel == 9 || kernel == 10) ? 128 : 256;
}
void gemm_bf16_allow_320(int on) { g_allow_320 = on; }
void gemm_bf16_set_shared_gpu(int on) { g_shared_gpu = on; }
int gemm_bf16_shared_gpu() { return g_shared_gpu; }

// Number of K slices a split-K launch actually produces (matches launch_variant)
int gemm_bf16_num_slices(int K, int split_k) {
  const int nk = K / BK;
  const int split = split_k > 1 ? split_k : 1;
  const int per = (int)ceil_div(nk, split);
  return (int)ceil_div(nk, per);
}

static void dispatch_gemm_bf16(const Bf16GemmArgs& g, const EpiParams& ep, int mode, hipStream_t s) {
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

// ---- tail balancing (round 6).  A persistent launch of T tiles on 256 workgroups runs ceil(T / 256) rounds: 591 tiles (M = 50432, N = 768, 256-row
// tiles) are 2.31 rounds run as 3, 2364 (N = 3072) 9.23 run as 10, ViT-L/16's 788 (N = 1024) 3.08 run as 4, DeepViT cfg4's 260 1.02 run as 2.  Here the rows
// of the whole rounds go out as one persistent launch (every workgroup the same number of tiles) and the ROWS OF THE LAST PARTIAL ROUND as a second launch of
// SMALLER tiles, one per workgroup (216 tiles of 192 x 128 or 240 of 128 x 128 instead of 79 / 60 of 256 x 256): every CU has work in the tail and the tail
// costs ~0.4 of a round instead of a whole one.  A sub-launch is the same GEMM on a row range (operand / output / residual / aux pointers advanced by the
// first row), so every fused epilogue applies unchanged and -- unlike a split along K (stream-K) -- no fp32 partial leaves the chip and every output element
// is accumulated in the same K order by one wave: results are bit-identical to the single launch (the per-tile column sums of EPI_GELU_BWD are summed over
// other row groups: same value up to the order of a fixed-order fp32 sum).  Why not stream-K: the tail's partials would be 256 KiB of fp32 per (tile, slice)
// -- at K = 768 as many bytes as the operands of the slice -- written, re-read and reduced by a fixup pass; measured alternatives in
// profiles/r6/gemm_tail_balancing_r6.md.
static bool tail_split_plan(const Bf16GemmArgs& g, int mode, int main_variant, int* rows_main) {
  if (g.tail <= 0 || g.split_k > 1 || g_shared_gpu || g.shared_gpu) return false;
  // (row-remapping / partial epilogues: one launch.  EPI_BIAS_GELU too: only the 256-row tile has LDS to spare for the GELU table, the small tiles evaluate
  //  the polynomial form -- the tail rows would get another rounding of gelu than the rows in front of them)
  if (!(mode == EPI_STORE || mode == EPI_BIAS_RESID || mode == EPI_GELU_BWD)) return false;
  if (!(main_variant == 13 || main_variant == 11)) return false;
  const int bm = main_variant == 11 ? 320 : 256;
  const int64_t tiles_n = ceil_div(g.N, 256), tiles_m = ceil_div(g.M, bm), total = tiles_m * tiles_n, grid = 256;
  if (total <= grid || total % grid == 0) return false;
  const int64_t main_tm = (total / grid) * grid / tiles_n;   // row tiles the whole rounds cover
  if (main_tm <= 0 || main_tm >= tiles_m) return false;
  *rows_main = (int)(main_tm * bm);
  return true;
}
static void dispatch_gemm_bf16_tail(const Bf16GemmArgs& g, const EpiParams& ep, int mode, hipStream_t s) {
  int r0 = 0;
  if (!tail_split_plan(g, mode, g.kernel & 15, &r0)) { Bf16GemmArgs g1 = g; g1.tail = 0; dispatch_gemm_bf16(g1, ep, mode, s); return; }
  Bf16GemmArgs gm = g, gt = g;
  EpiParams em = ep, et = ep;
  gm.tail = gt.tail = 0;
  gm.M = em.M = r0;
  gt.M = et.M = g.M - r0;
  gt.kernel = g.tail;
  gt.A = g.A + (int64_t)r0 * g.lda;
  const int osz = (mode == EPI_BIAS_RESID) ? 4 : 2;
  auto adv = [](const void* p, int64_t bytes) -> void* { return p ? (void*)((char*)p + bytes) : nullptr; };
  et.out = adv(ep.out, (int64_t)r0 * ep.ldo * osz);
  et.out2 = adv(ep.out2, (int64_t)r0 * ep.ldo2 * 2);
  et.resid = (const float*)adv(ep.resid, (int64_t)r0 * ep.ldr * 4);
  et.aux = adv(ep.aux, (int64_t)r0 * ep.ldaux * 2);
  // per-tile column sums: the main launch writes rows [0, 2 r0 / 256) (one per 128-row wave row; 320-row tiles: per 160), the tail's rows follow
  // (<= one per 96 rows: the caller's buffer has ceil(M / 96) rows)
  const int main_cs_rows = (g.kernel & 15) == 11 ? 2 * (r0 / 320) : 2 * (r0 / 256);
  et.colsum = ep.colsum ? ep.colsum + (int64_t)main_cs_rows * ep.ldcs : nullptr;
  // row tiles walked from the last to the first (an A operand larger than the memory-side cache, just written front to back): the tail rows are the last
  if (g.reverse_m) { dispatch_gemm_bf16(gt, et, mode, s); dispatch_gemm_bf16(gm, em, mode, s); }
  else { dispatch_gemm_bf16(gm, em, mode, s); dispatch_gemm_bf16(gt, et, mode, s); }
}

// ---- per-shape variant selection by measurement.  The variants differ by a few percent per (shape, epilogue) and the ranking
// moves with the board's clocks, so with kernel = 0 (automatic) the first launch of each (mode, M, N, K) times the candidates on
// the caller's stream with the caller's operands (every fused epilogue is a pure function of its inputs -- the launch is
// repeatable as long as the output does not alias the residual) and caches the winner for the process.  All variants accumulate
// every output element in the same K order, so the choice does not change results.  VITX_GEMM_AUTOTUNE=0 falls back to the
// static rule gemm_bf16_pick().
static std::mutex g_tune_mu;
static std::map<std::array<int64_t, 6>, int> g_tuned;
// VITX_GEMM_TAIL=1: the per-shape measurement also times the tail-balanced forms (same results either way).  OFF by default: in isolation the split
// wins 2-5 % on the launches it applies to (out-proj + residual 117.9 -> 112.7 us, ViT-L/16's N = 1024 launches 174 -> 168, 454 -> 436 us), but inside the
// training step every extra launch boundary of the input-gradient chain is a gap in which the low-priority weight-gradient stream places its persistent
// workgroups, and the tail then queues behind them: ViT-B/16 35.24 -> 35.36 ms, ViT-L/16 109.5 -> 111.2 ms, DeepViT / CaiT neutral
// (profiles/r6/gemm_tail_balancing_r6.md, same box, alternating runs).
static int tail_enabled() {
  static int on = -1;
  if (on < 0) { const char* v = vitx_env("VITX_GEMM_TAIL"); on = (v && atoi(v) != 0) ? 1 : 0; }
  return on;
}
static int autotune_enabled() {
  static int on = -1;
  if (on < 0) { const char* v = vitx_env("VITX_GEMM_AUTOTUNE"); on = (v && atoi(v) == 0) ? 0 : 1; }
  return on;
}
void launch_gemm_bf16(const Bf16GemmArgs& g0, const EpiParams& ep, int mode, hipStream_t s) {
  // small batches: a few hundred token rows give the 256x256 tiles less than two per CU, where the smaller tiles can win by a lot
  const bool few_tiles = ceil_div(g0.M, 256) * ceil_div(g0.N, 256) < 512;
  const double work = (double)g0.M * g0.N * g0.K;
  const bool tunable = g0.kernel == 0 && autotune_enabled() && (g0.N % 256 == 0 || g0.N > 512) && (work >= 2.0e9 || (few_tiles && work >= 1.0e8)) &&
                       !(mode == EPI_BIAS_RESID && ep.out == (void*)ep.resid);
  if (!tunable) { dispatch_gemm_bf16_tail(g0, ep, mode, s); return; }
  const std::array<int64_t, 6> key = {mode, g0.M, g0.N, g0.K, g0.split_k > 1 ? g0.split_k : 1, (ep.scale != nullptr) + 2 * (g_shared_gpu | (g0.shared_gpu != 0))};
  int best = -1;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) best = it->second;
  }
  Bf16GemmArgs g = g0;
  if (best < 0) {
    // At most four candidates that differ by more than the measurement noise: the pipelined persistent kernel with 256- and 320-row tiles (their
    // ranking is the tile count against 256 CUs: 768-wide outputs take 320 rows, wider ones 256) and the two small tiles for launches with few
    // row tiles; beside collectives (data parallel) the one-tile-per-workgroup forms 2 / 5 instead of the persistent ones.
    static const int cand[] = {13, 2, 11, 5, 3, 1};
    // 256x128 / 128x128 tiles only compete when 256x256 tiles cannot give every CU two of them (token subsets: MAE's encoder
    // sees 49 of 196 patches, M = 12544 -> 147 tiles for a 768-wide output)
    const bool small_m = ceil_div(g0.M, 256) * ceil_div(g0.N, 256) < 512;
    // each candidate: one warm-up, then NREP individually timed launches; the candidate's time is the FASTEST of them (a launch can
    // only be delayed by interference, never sped up), so one hiccup does not hand the shape to a slower variant for the whole process
    constexpr int NREP = 5;
    hipEvent_t ev[NREP + 1];
    for (auto& x : ev) (void)hipEventCreate(&x);
    float best_ms = 1e30f;
    best = gemm_bf16_pick(g0.M, g0.N);
    auto measure = [&](int c, int tail) {
      g.kernel = c; g.tail = tail;
      dispatch_gemm_bf16_tail(g, ep, mode, s);   // warm-up (first-use attribute setup, instruction cache)
      (void)hipEventRecord(ev[0], s);
      for (int r = 0; r < NREP; ++r) {
        dispatch_gemm_bf16_tail(g, ep, mode, s);
        (void)hipEventRecord(ev[r + 1], s);
      }
      if (hipEventSynchronize(ev[NREP]) != hipSuccess) return;
      float fastest = 1e30f;
      for (int r = 0; r < NREP; ++r) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[r], ev[r + 1]) == hipSuccess && ms < fastest) fastest = ms;
      }
      if (fastest < best_ms * 0.985f) { best_ms = fastest; best = c + 32 * tail; }   // a later candidate must win by more than the noise: the same pick run after run
    };
    for (int c : cand) {
      const bool is320 = c == 5 || c == 11;
      if (is320 && !g_allow_320) continue;
      if ((c == 1 || c == 3) && !small_m) continue;
      if (!(g_shared_gpu || g0.shared_gpu) && (c == 2 || c == 5)) continue;   // beside collectives the pipelined kernel runs one tile per workgroup too (launch_pipe) and competes with 2 / 5
      measure(c, 0);
    }
    // (round 6) tail balancing: the whole rounds on 256-row tiles + the rows of the last partial round on small tiles (dispatch_gemm_bf16_tail)
    if (tail_enabled()) {
      int r0 = 0;
      g.kernel = 13; g.tail = 1;
      if (tail_split_plan(g, mode, 13, &r0))
        for (int t : {10, 1, 3}) measure(13, t);
    }
    g.tail = 0;
    for (auto& x : ev) (void)hipEventDestroy(x);
    if (vitx_env("VITX_GEMM_AUTOTUNE_LOG"))
      fprintf(stderr, "[vitx] gemm autotune: mode %d M %d N %d K %d split %d -> variant %d tail %d (%.4f ms)\n", mode, g0.M, g0.N, g0.K, g0.split_k, best & 31,
              best >> 5, best_ms);
    // The candidates have different tile heights and each stores its per-tile column sums (EPI_GELU_BWD) with '=' into rows the caller
    // zeroed ONCE: rows a taller-tiled winner does not write would keep what a shorter-tiled candidate left there, and the reduction behind
    // the launch adds every row (the fc1 bias gradient of the first step of a process was wrong by that much).  Clear them again.
    if (ep.colsum != nullptr) (void)hipMemsetAsync(ep.colsum, 0, (size_t)ceil_div(g0.M, 96) * ep.ldcs * sizeof(float), s);
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = best;
  }
  g.kernel = best & 31;
  g.tail = best >> 5;
  dispatch_gemm_bf16_tail(g, ep, mode, s);
}

void launch_gemm_bf16_persistent_lockstep(int bm, int mode, const Bf16GemmArgs& g0, const EpiParams& ep, hipStream_t s) {
  Bf16GemmArgs g = g0;
  g.kernel = (bm == 320 ? 7 : 6) | 512;   // persistent lockstep variant, LDS-staged epilogue
  dispatch_gemm_bf16(g, ep, mode, s);
}
