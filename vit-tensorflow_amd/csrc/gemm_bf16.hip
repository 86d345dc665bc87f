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
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "kernels.h"

static int g_allow_320 = 1;
static int g_shared_gpu = 0;
int gemm_bf16_pick(int M, int N);

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

constexpr int BK = 64;

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int V> using ic = std::integral_constant<int, V>;

// logical tile index -> (tile_m, tile_n).  Wide outputs (>= 8 column tiles) are walked in bands of 4 row tiles, column-major inside a
// band, so the ~32 consecutive tiles an XCD works on at any time form a 4 x 8 block: 12 operand panels in flight instead of 2.7 + 12
// (PMC: the row-major order re-fetched the A panel of fc1 5.5x and of qkv 3.7x through the XCD's 4 MiB L2; 8192^3 +12 %).
__device__ __forceinline__ void decode_tile_fwd(int L, int tiles_m, int tiles_n, int gm, int& tm, int& tn);
// gm < 0: the same walk with the row tiles taken from the LAST to the first.  For a GEMM whose A operand was just written, front to back,
// by the previous kernel and is larger than the 256 MB memory-side cache (act for fc2, d(hpre) for the fc1 dgrad: 310 MB): the cache holds
// the most recently written rows, and a reader that starts at row 0 misses, allocates, and evicts exactly the rows it needs next; starting
// at the end it hits on everything that is still there.
__device__ __forceinline__ void decode_tile(int L, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
  if (gm < 0) { decode_tile_fwd(L, tiles_m, tiles_n, -gm, tm, tn); tm = tiles_m - 1 - tm; return; }
  decode_tile_fwd(L, tiles_m, tiles_n, gm, tm, tn);
}
__device__ __forceinline__ void decode_tile_fwd(int L, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
  if (gm == 1) { tm = L / tiles_n; tn = L - tm * tiles_n; return; }
  const int group = gm * tiles_n;
  const int gid = L / group, first = gid * gm;
  const int gsz = min(tiles_m - first, gm);
  const int w = L - gid * group;
  tn = w / gsz;
  tm = first + (w - tn * gsz);
}

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
          if (has_bias) {
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
  const unsigned gx = (SCHED == 2 && zs == 1) ? (unsigned)std::min(tiles_m * tiles_n, 256) : (unsigned)(tiles_m * tiles_n);
  dim3 grid(gx, (unsigned)zs), block(WM * WN * 64);
  hipLaunchKernelGGL(kern, grid, block, SMEM, s, g, ep, tiles_m, tiles_n, per);
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined persistent NT kernel (the default for N >= 256).  Same tile / wave decomposition, LDS image and swizzle as
// gemm_bf16_nt_kernel, but
//   * the MFMA fragments are double-buffered in registers: the ds_reads of k-step s+1 are issued before the MFMAs of k-step s,
//     so the LDS latency (8 exposed lgkmcnt(0) waits per K-tile in the plain loop, taken by both waves of a SIMD at the same time)
//     disappears from the critical path; PMC on the plain loop: 38 % of wave cycles parked in s_waitcnt, MFMA pipe 54 % busy;
//   * the K-tile hand-over (vmcnt(0) + barrier + DMA issue for the tile after next + first fragments of the next tile) sits in
//     front of the LAST k-step's MFMAs instead of between two tiles;
//   * workgroups are persistent and their K-tile stream runs across output tiles (the next tile's first K-tile is in LDS before
//     the epilogue of the current one starts).
template <int BM, int BN, int WM, int WN, int MODE, int PAT>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_nt_pipe_kernel(Bf16GemmArgs g, EpiParams ep, int tiles_m, int tiles_n,
                                                                         int kt_per_split) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  constexpr int SROW = BN + 4;                     // epilogue staging row (floats); +16 B keeps ds_write_b128 groups conflict-free
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
  static_assert(BN == 256 && 2 * 32 * BN * 4 <= STAGE, "two 32-row epilogue staging areas must fit one pipeline buffer");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int logical0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int total_tiles = tiles_m * tiles_n;
  if (logical0 >= total_tiles) return;
  const int z = blockIdx.y;
  const int kt0 = z * kt_per_split;
  const int nk = min(kt_per_split, g.K / BK - kt0);
  const bool persistent = gridDim.y == 1;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  if (g.phase > 0 && persistent) {   // de-phase the workgroups of an XCD: their store-heavy epilogues stop coinciding
    const int ph = (bid >> 3) & 7;
    for (int i = 0; i < ph * g.phase; ++i) __builtin_amdgcn_s_sleep(64);
  }

  // per-lane DMA source offsets in BYTES, unsigned: address = uniform 64-bit base (SGPR pair) + zero-extended 32-bit lane offset,
  // which selects the saddr form of global_load_lds (one address dword per lane, no per-piece VALU address arithmetic)
  uint32_t offA[A_INSTR], offB[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    offA[i] = (uint32_t)(row * (int)g.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    offB[i] = (uint32_t)(row * (int)g.ldb + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2u;
  }

  const int gm = ((tiles_n >= 8 && g.stagger != 8) ? 4 : 1) * (g.reverse_m ? -1 : 1);   // (stagger 8: row-major order, A/B switch of the micro-benchmark)
  // ---- issue cursor over this workgroup's K-tile stream (tile, kt); LDS buffer of stream item i = i & 1
  int i_logical = logical0, i_k = 0, issued = 0;
  bool i_more = nk > 0;
  // DMA addressing: buffer_load ... lds with one resource per operand (SGPRs), the tile / K offset in the scalar offset and a
  // loop-invariant 32-bit lane offset -- no per-piece VALU address arithmetic and one address dword per lane instead of two
  // (the flat global_load_lds form needs a 64-bit address per lane).  Operands are < 4 GiB (checked by the launcher).
  // The pieces are issued from inline asm (vitx_dma16, common.h): as builtins the compiler drained them with vmcnt(0) in front of the
  // next k-step's fragment reads.
  const i32x4 rsA = vitx_make_rsrc(g.A), rsB = vitx_make_rsrc(g.B);
  const uint32_t lds_w = vitx_lds_addr(smem) + (uint32_t)wave * 1024u;   // this wave's 1-KiB slot inside an 8-KiB piece row
  uint32_t a_soff = 0, b_soff = 0;       // byte offset of the cursor tile's first K-tile inside A / B
  auto i_set_tile = [&]() {
    int tm, tn;
    decode_tile(i_logical, tiles_m, tiles_n, gm, tm, tn);
    a_soff = (uint32_t)(((int64_t)tm * BM * g.lda + (int64_t)kt0 * BK) * 2);
    b_soff = (uint32_t)(((int64_t)tn * BN * g.ldb + (int64_t)kt0 * BK) * 2);
  };
  i_set_tile();
  constexpr int P = A_INSTR + B_INSTR;   // DMA pieces (1 KiB each) per K-tile per wave
  // where the P pieces of K-tile it+2 are issued: k-step 3 of tile it (after the hand-over), k-steps 0 and 1 of tile it+1.
  // 64 pieces issued by 8 waves at the same moment queue up behind one address unit (~50 cycles each, all waves blocked);
  // spread over the tile each one costs ~18 cycles and hides under an MFMA.
  constexpr int N3 = PAT == 0 ? P : (PAT == 1 ? (P + 1) / 2 : (P + 2) / 3);
  constexpr int N0 = PAT == 0 ? 0 : (PAT == 1 ? P / 2 : (P + 1) / 3);
  constexpr int N1 = P - N3 - N0;
  bool pending = false;                  // pieces of the cursor's K-tile still to be issued
  const int xp = g.stagger;   // timing experiments only (results are wrong): 1 = no DMA wait, 2 = no DMA issue in the K loop
  auto issue_piece = [&](uint32_t base, auto p_c) {   // base = LDS byte offset of the target stage
    constexpr int p = decltype(p_c)::value;
    if constexpr (p < A_INSTR) vitx_dma16(rsA, lds_w + base + p * NW * 1024, offA[p], a_soff + i_k * (BK * 2));
    else vitx_dma16(rsB, lds_w + base + A_BYTES + (p - A_INSTR) * NW * 1024, offB[p - A_INSTR], b_soff + i_k * (BK * 2));
  };
  auto i_advance = [&]() {
    ++issued;
    if (++i_k == nk) {
      i_k = 0;
      i_logical += nwg;
      i_more = persistent && i_logical < total_tiles;
      if (i_more) i_set_tile();
    }
  };
  auto issue = [&]() {
    if (!i_more) return;
    const uint32_t base = (issued & 1) * STAGE;
    static_for<P>([&](auto p_c) { issue_piece(base, p_c); });
    i_advance();
  };

  // fragment addressing: row = tile row of this lane (lane&31), chunk = (ks*2 + (lane>>5)) ^ swz(row)
  const int sw = ((lane & 31) >> 1) & 7;
  const int a_row_byte = (wm * WTM + (lane & 31)) * 128;
  const int b_row_byte = A_BYTES + (wn * WTN + (lane & 31)) * 128;
  const int khalf = lane >> 5;
  bf16x8 fa[2][MT], fb[2][NT];
  auto load_frags = [&](bf16x8(&af)[MT], bf16x8(&bfr)[NT], const char* base, int ks) {
    const int cb = ((ks * 2 + khalf) ^ sw) << 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8*)(base + a_row_byte + i * 32 * 128 + cb);
#pragma unroll
    for (int j = 0; j < NT; ++j) bfr[j] = *(const bf16x8*)(base + b_row_byte + j * 32 * 128 + cb);
  };
  constexpr int Q = MT * NT;             // MFMAs per k-step per wave
  f32x16 acc[MT][NT];
  auto mfma_range = [&](auto cur_c, auto first_c, auto last_c) {   // MFMAs [first, last) of a k-step, fragments set `cur`
    constexpr int CUR = decltype(cur_c)::value, FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
    static_for<(LAST > FIRST ? LAST - FIRST : 0)>([&](auto d) {
      constexpr int idx = FIRST + decltype(d)::value, i = idx / NT, j = idx % NT;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][j], fa[CUR][i], acc[i][j], 0, 0, 0);
    });
  };
  // The wait is the BUILTIN, not asm: the compiler models it, so its own scoreboard is empty from here on.  With an asm wait it kept the
  // epilogue's global loads "pending" around the whole K loop and protected the registers they had written with `s_waitcnt vmcnt(1)` /
  // `vmcnt(0)` in front of the first fragment reads of every K-tile -- which, at run time, waited for the DMA pieces issued a few MFMAs earlier.
  auto handover = [&]() {   // every wave's reads of the older buffer are in registers, the younger buffer has landed
    if (xp & 1) __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  const bool has_bias = ep.bias != nullptr, has_scale = ep.scale != nullptr;
  const int64_t out_off = (int64_t)z * ep.partial_stride;

  issue();      // stream items 0 and 1
  issue();
  handover();
  load_frags(fa[0], fb[0], smem, 0);
  int it = 0;   // consumed K-tile counter of the stream
  int tile_idx = 0;
  constexpr bool kStamps = false;   // cycle stamps of the tile phases (diagnostic build only: the stores leave VMEM state pending across the K loop)
  auto stamp = [&](int k) {
    if constexpr (kStamps)
      if (g.stamps && tid == 0 && bid < 256 && tile_idx < 16) g.stamps[((int64_t)bid * 16 + tile_idx) * 4 + k] = __builtin_readcyclecounter();
  };
  // (the ONLY back edge of this loop runs through handover(): on any other path the compiler's scoreboard would carry the epilogue's
  //  bias / residual loads into the K loop as "pending" and protect their registers with vmcnt waits there -- see handover())
  for (int logical = logical0;; logical += nwg, ++tile_idx) {
    int tile_m, tile_n;
    decode_tile(logical, tiles_m, tiles_n, gm, tile_m, tile_n);
    const bool has_next = persistent && logical + nwg < total_tiles;
    stamp(0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
      const char* base = smem + (it & 1) * STAGE;
      static_for<BK / 16>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value, CUR = ks & 1;
        // Order is pinned with sched_barrier(0): two MFMAs, then the ds_reads of the NEXT k-step, then the remaining MFMAs with
        // this k-step's share of the DMA pieces between them (the reads are >= Q-2 MFMAs old when their consumer arrives; left
        // alone, the scheduler sinks them to just before use).
        constexpr int NP = ks == 3 ? N3 : (ks == 0 ? N0 : (ks == 1 ? N1 : 0));   // DMA pieces issued in this k-step
        constexpr int FP = ks == 3 ? 0 : (ks == 0 ? N3 : N3 + N0);              // first of them
        if constexpr (ks + 1 < BK / 16) {
          mfma_range(ic<CUR>{}, ic<0>{}, ic<2>{});
          __builtin_amdgcn_sched_barrier(0);
          load_frags(fa[CUR ^ 1], fb[CUR ^ 1], base, ks + 1);
        } else {
          // K-tile hand-over in front of the last k-step's MFMAs (ONE instruction stream for every case -- branching the MFMA
          // sequence makes the allocator copy accumulators): K-tile it+1 has landed, buffer it&1 is fully read by every wave.
          handover();
          mfma_range(ic<CUR>{}, ic<0>{}, ic<2>{});
          __builtin_amdgcn_sched_barrier(0);
          // first fragments of the next K-tile; at the last K-tile of an output tile they are loaded AFTER the epilogue instead
          // (kept live across it they cost the 320-row variants their last free registers)
          if (kt + 1 < nk) load_frags(fa[0], fb[0], smem + ((it + 1) & 1) * STAGE, 0);
          // K-tile it+2 goes into the buffer just released; at the last K-tile of an output tile the refill is deferred until
          // after the epilogue, which stages through that buffer.
          pending = i_more && kt + 1 < nk && !(xp & 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NP > 0) {
          const uint32_t ibase = (issued & 1) * STAGE;
          static_for<NP>([&](auto d_c) {
            constexpr int d = decltype(d_c)::value;
            mfma_range(ic<CUR>{}, ic<(2 + d < Q ? 2 + d : Q)>{}, ic<(3 + d < Q ? 3 + d : Q)>{});
            if (pending) issue_piece(ibase, ic<FP + d>{});
            __builtin_amdgcn_sched_barrier(0);
          });
          mfma_range(ic<CUR>{}, ic<(2 + NP < Q ? 2 + NP : Q)>{}, ic<Q>{});
          if constexpr (FP + NP == P) {
            if (pending) i_advance();
            pending = false;
          }
        } else {
          mfma_range(ic<CUR>{}, ic<2>{}, ic<Q>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      ++it;
    }

    // ---- epilogue: 32 output rows per round through the buffer of the last K-tile (its refill is deferred until after the epilogue)
    stamp(1);
    {
      const bool interior = epilogue_fast_ok(ep, MODE) && (tile_m + 1) * BM <= ep.M && (tile_n + 1) * BN <= ep.N;
      float* st = (float*)(smem + ((it + 1) & 1) * STAGE);
      const int gcol = tile_n * BN + lane * 4;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (interior && has_bias) b4 = *(const float4*)(ep.bias + gcol);
      if (interior && has_scale) s4 = *(const float4*)(ep.scale + gcol);
      constexpr int RPW = 32 / NW;   // rows per wave per round (one 1-KiB row per wave instruction)
      float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);   // EPI_GELU_BWD: column sums of what this lane stores (fused bias gradient)
      auto cs_add = [&](float4 r) { if (MODE == EPI_GELU_BWD) { cs.x += r.x; cs.y += r.y; cs.z += r.z; cs.w += r.w; } };
      // Two 32-row staging areas ([32][BN] fp32 = 32 KiB each, 16-B chunks swizzled chunk ^= row & 7 instead of padded rows) used
      // alternately: ONE barrier per round -- the accumulator rows of round R+1 are written while round R is still being read and
      // stored (the round trip  barrier - LDS write - barrier - LDS read - global store  was the epilogue's critical path, not HBM).
      // Reuse of an area two rounds later is ordered by the barrier in between (every wave waits for its own reads first).
      // bf16 outputs (WIDE): a lane takes EIGHT consecutive columns of a row (two staged 16-B chunks -> one 16-B global access per
      // output, two rows per wave instruction): half the store instructions for the same bytes.  Even chunks of a staged row live
      // in its first 512 B, odd chunks in the second, so both reads of a 16-lane group stay conflict-free.
      // (256-row tiles only: in the 320-row variants the second code path costs the registers the accumulators need -- 44-116 B of scratch;
      //  same-box A/B on fc1-shaped launches, profiles/r2/epilogue_wide_ab_r2d.log: +1.0..2.5 % plain store, +1 % GELU, +2.5 % GELU VJP)
      constexpr bool WIDE = BM == 256 && (MODE == EPI_STORE || MODE == EPI_BIAS_GELU || MODE == EPI_GELU_BWD);
      const bool wide = WIDE && interior && ep.wide_ok;
      float4 cs2 = make_float4(0.f, 0.f, 0.f, 0.f);   // WIDE column sums: columns 4..7 of the lane's eight
      const int cc = lane & 31, gcol8 = tile_n * BN + cc * 8;
      float4 b8lo = make_float4(0.f, 0.f, 0.f, 0.f), b8hi = b8lo;
      if (wide && has_bias && MODE != EPI_GELU_BWD) { b8lo = *(const float4*)(ep.bias + gcol8); b8hi = *(const float4*)(ep.bias + gcol8 + 4); }
#pragma clang loop unroll(full)
      for (int R = 0; R < BM / 32; ++R) {
        const int wm_r = (R * 32) / WTM, i0 = ((R * 32) % WTM) / 32;
        float* sr = st + (R & 1) * (32 * BN);
        if (wm == wm_r) {
          const int m = lane & 31;
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int chunk = (wn * WTN + j * 32 + 8 * q + 4 * khalf) >> 2;
              const int pc = WIDE ? ((chunk >> 1) | ((chunk & 1) << 5)) : chunk;
              *(float4*)(sr + m * BN + ((pc ^ (m & 7)) << 2)) =
                  make_float4(acc[i0][j][4 * q], acc[i0][j][4 * q + 1], acc[i0][j][4 * q + 2], acc[i0][j][4 * q + 3]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int grow0 = tile_m * BM + R * 32;
        if constexpr (WIDE) {
          if (wide) {
            constexpr int RPW2 = 16 / NW;    // two rows per wave instruction, 16 instructions per 32-row round
            float4 lo[RPW2], hi[RPW2];
            bf16x8 x[RPW2];
#pragma unroll
            for (int k = 0; k < RPW2; ++k) {
              const int r = (k * NW + wave) * 2 + (lane >> 5);
              const int pc = cc ^ (r & 7);
              lo[k] = *(const float4*)(sr + r * BN + (pc << 2));
              hi[k] = *(const float4*)(sr + r * BN + ((pc + 32) << 2));
              x[k] = epilogue_wide_load<MODE>(ep, grow0 + r, gcol8);
            }
#pragma unroll
            for (int k = 0; k < RPW2; ++k) {
              const int r = (k * NW + wave) * 2 + (lane >> 5);
              if (has_bias) epilogue_wide8<MODE, true>(ep, grow0 + r, gcol8, lo[k], hi[k], b8lo, b8hi, x[k], out_off);
              else epilogue_wide8<MODE, false>(ep, grow0 + r, gcol8, lo[k], hi[k], b8lo, b8hi, x[k], out_off);
              if (MODE == EPI_GELU_BWD) {
                cs.x += lo[k].x; cs.y += lo[k].y; cs.z += lo[k].z; cs.w += lo[k].w;
                cs2.x += hi[k].x; cs2.y += hi[k].y; cs2.z += hi[k].z; cs2.w += hi[k].w;
              }
            }
            continue;
          }
        }
        float4 v[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
          const int r = k * NW + wave;
          const int pc = WIDE ? (((lane >> 1) | ((lane & 1) << 5)) ^ (r & 7)) : (lane ^ (r & 7));
          v[k] = *(const float4*)(sr + r * BN + (pc << 2));
        }
        if (interior) {
          float4 x[RPW];
#pragma unroll
          for (int k = 0; k < RPW; ++k) x[k] = epilogue_fast_load<MODE, bf16_t>(ep, grow0 + k * NW + wave, gcol);
          if (has_bias) {
#pragma unroll
            for (int k = 0; k < RPW; ++k) cs_add(epilogue_fast4<MODE, bf16_t, true, false>(ep, grow0 + k * NW + wave, gcol, v[k], b4, s4, x[k], out_off));
          } else {
#pragma unroll
            for (int k = 0; k < RPW; ++k) cs_add(epilogue_fast4<MODE, bf16_t, false, false>(ep, grow0 + k * NW + wave, gcol, v[k], b4, s4, x[k], out_off));
          }
        } else {
#pragma unroll
          for (int k = 0; k < RPW; ++k) cs_add(epilogue_apply4<MODE, bf16_t>(ep, grow0 + k * NW + wave, gcol, v[k], out_off));
        }
      }
      if (MODE == EPI_GELU_BWD && ep.colsum != nullptr) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // per-tile column sums: partial rows through the staging buffer (one per wave; two per wave in the eight-column form)
        const int nrows = wide ? 2 * NW : NW;
        if (wide) {
          *(float4*)(st + (wave * 2 + (lane >> 5)) * BN + cc * 8) = cs;
          *(float4*)(st + (wave * 2 + (lane >> 5)) * BN + cc * 8 + 4) = cs2;
        } else {
          *(float4*)(st + wave * BN + lane * 4) = cs;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid < BN) {
          float a = 0.f;
          for (int w = 0; w < nrows; ++w) a += st[w * BN + tid];
          const int c = tile_n * BN + tid;
          if (c < ep.N) ep.colsum[(int64_t)tile_m * ep.ldcs + c] = a;
        }
      }
    }
    stamp(2);
    // modelled wait on EVERY path out of the epilogue (the structurizer routes the `break` through the block that is also the loop latch, so a
    // wait on the continue path alone leaves the epilogue's loads pending at the loop header in the compiler's view)
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the epilogue's stores have left the wave
    if (!has_next) break;
    handover();                                   // staging reads done everywhere
    issue();                                      // deferred refill of the staging buffer: stream item it+1
    load_frags(fa[0], fb[0], smem + (it & 1) * STAGE, 0);
    stamp(3);
  }
}

template <int BM, int BN, int WM, int WN, int MODE, int PAT = 0>
void launch_pipe(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  constexpr int SMEM = 2 * (BM + BN) * BK * 2;
  // buffer-addressed DMA: 31-bit byte offsets inside each operand; larger operands take the flat-addressed persistent kernel
  if (((int64_t)ceil_div(g.M, BM) * BM * g.lda + g.K) * 2 >= (1LL << 31) || ((int64_t)ceil_div(g.N, BN) * BN * g.ldb + g.K) * 2 >= (1LL << 31)) {
    launch_variant<BM, BN, WM, WN, MODE, true, 2>(g, ep, s);
    return;
  }
  auto kern = gemm_bf16_nt_pipe_kernel<BM, BN, WM, WN, MODE, PAT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  const int tiles_m = (int)ceil_div(g.M, BM), tiles_n = (int)ceil_div(g.N, BN);
  const int nk = g.K / BK;
  const int split = g.split_k > 1 ? g.split_k : 1;
  const int per = (int)ceil_div(nk, split);
  const int zs = (int)ceil_div(nk, per);
  static const int phase_env = [] { const char* v = getenv("VITX_GEMM_PHASE"); return v ? atoi(v) : 0; }();
  Bf16GemmArgs gp = g;
  if (gp.phase == 0) gp.phase = phase_env;
  static const int grid_cap = [] { const char* v = getenv("VITX_GEMM_GRID"); return v ? atoi(v) : 256; }();   // experiment: fewer persistent workgroups
  const unsigned gx = zs == 1 ? (unsigned)std::min(tiles_m * tiles_n, grid_cap) : (unsigned)(tiles_m * tiles_n);
  dim3 grid(gx, (unsigned)zs), block(WM * WN * 64);
  hipLaunchKernelGGL(kern, grid, block, SMEM, s, gp, ep, tiles_m, tiles_n, per);
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
  // 1-D grid over (K-slice, tile), K-slice major.  Workgroup ids go round-robin to the 8 XCDs, so XCD x is given a CONTIGUOUS
  // chunk of that list: (almost) all tiles of one K-slice run on one XCD at the same time, and the slice of X / dY they share is
  // fetched from HBM once into that XCD's L2 instead of once per XCD (PMC: 898 MB fetched per launch for 310 MB of operands
  // with the slice index on gridDim.y, where the linear id -- hence the XCD -- mixes slices).
  const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles = tiles_m * tiles_n;
  const int z = work / tiles;
  const int logical = work - z * tiles;
  const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
  const int nk_total = g.K / BK;
  const int kt0 = z * kt_per_split;
  const int nk = min(kt_per_split, nk_total - kt0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  // DMA: instruction i of this wave covers token rows (i*NW + wave)*RPI .. +RPI-1; lane -> (row, physical chunk)
  const bf16_t* Ag = g.A + (int64_t)kt0 * BK * g.lda + (int64_t)tile_m * BT;
  const bf16_t* Bg = g.B + (int64_t)kt0 * BK * g.ldb + (int64_t)tile_n * BT;
  // per-lane DMA source offsets in bytes (unsigned 32-bit: uniform 64-bit base + zero-extended lane offset)
  uint32_t offA[INSTR], offB[INSTR];
#pragma unroll
  for (int i = 0; i < INSTR; ++i) {
    const int row = (i * NW + wave) * RPI + lane / LPR;
    const int pc = lane % LPR;
    const int c = ((((pc >> 1) ^ (2 * (row & 3))) << 1) | (pc & 1));   // logical 16-B chunk stored at physical chunk pc
    offA[i] = (uint32_t)(row * (int)g.lda + c * 8) * 2u;
    offB[i] = (uint32_t)(row * (int)g.ldb + c * 8) * 2u;
  }
  constexpr int P = 2 * INSTR;           // DMA pieces (1 KiB) per K-tile per wave: A pieces, then B pieces
  constexpr int Q = MT * NT;             // MFMAs per k-step per wave
  // pieces of K-tile kt+2 are issued in k-step 3 of tile kt (after the hand-over) and k-steps 0, 1 of tile kt+1 -- see the NT kernel
  constexpr int N3 = (P + 2) / 3, N0 = (P + 1) / 3, N1 = P - N3 - N0;
  // buffer-addressed DMA (see the NT pipe kernel): resource per operand, K-tile offset in the scalar offset
  // issued from inline asm (vitx_dma16, common.h): as builtins the compiler put `s_waitcnt vmcnt(0)` in front of EVERY k-step's transpose reads
  const i32x4 rsA = vitx_make_rsrc(Ag), rsB = vitx_make_rsrc(Bg);
  const uint32_t lds_w = vitx_lds_addr(smem) + (uint32_t)wave * 1024u;
  const uint32_t a_kstep = (uint32_t)(BK * g.lda * 2), b_kstep = (uint32_t)(BK * g.ldb * 2);   // bytes per K-tile (64 token rows)
  auto issue_piece = [&](int buf, int kt, auto p_c) {
    constexpr int p = decltype(p_c)::value;
    const uint32_t base = lds_w + (uint32_t)buf * STAGE;
    if constexpr (p < INSTR) vitx_dma16(rsA, base + p * NW * 1024, offA[p], (uint32_t)kt * a_kstep);
    else vitx_dma16(rsB, base + OP_BYTES + (p - INSTR) * NW * 1024, offB[p - INSTR], (uint32_t)kt * b_kstep);
  };
  auto stage = [&](int buf, int kt) { static_for<P>([&](auto p_c) { issue_piece(buf, kt, p_c); }); };

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
  bf16x8 fa[2][MT], fb[2][NT];                              // register double-buffered fragments (see the NT kernel)
  auto load_frags = [&](bf16x8(&af)[MT], bf16x8(&bfr)[NT], const char* base, int ks) {
    const int m0 = 16 * ks + 8 * khalf + trow, m1 = m0 + 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int fbyte = (wm * WTM + i * 32) * 2 + piece;
      af[i] = tr_frag(base + row_addr(m0, fbyte), base + row_addr(m1, fbyte));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int fbyte = (wn * WTN + j * 32) * 2 + piece;
      bfr[j] = tr_frag(base + OP_BYTES + row_addr(m0, fbyte), base + OP_BYTES + row_addr(m1, fbyte));
    }
  };
  auto mfma_range = [&](auto cur_c, auto first_c, auto last_c) {
    constexpr int CUR = decltype(cur_c)::value, FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
    static_for<(LAST > FIRST ? LAST - FIRST : 0)>([&](auto d) {
      constexpr int idx = FIRST + decltype(d)::value, i = idx / NT, j = idx % NT;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][j], fa[CUR][i], acc[i][j], 0, 0, 0);
    });
  };
  const int xp = g.stagger;   // timing experiments only (VITX_TN_XP; results are wrong): 1 = no DMA wait, 2 = no DMA issue in the K loop, 4 = no fragment reads in the K loop
  auto handover = [&]() {
    if (xp & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (nk > 0) stage(0, 0);
  if (nk > 1) stage(1, 1);
  handover();
  load_frags(fa[0], fb[0], smem, 0);
  bool pending = false;
  constexpr int QH = Q < 2 ? Q : 2;      // MFMAs issued ahead of the prefetch reads
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem + (kt & 1) * STAGE;
    static_for<BK / 16>([&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value, CUR = ks & 1;
      constexpr int NP = ks == 3 ? N3 : (ks == 0 ? N0 : (ks == 1 ? N1 : 0));
      constexpr int FP = ks == 3 ? 0 : (ks == 0 ? N3 : N3 + N0);
      if constexpr (ks + 1 < BK / 16) {
        mfma_range(ic<CUR>{}, ic<0>{}, ic<QH>{});
        __builtin_amdgcn_sched_barrier(0);
        if (!(xp & 4)) load_frags(fa[CUR ^ 1], fb[CUR ^ 1], base, ks + 1);
      } else {
        handover();                                                   // K-tile kt+1 landed; buffer kt&1 fully read by every wave
        mfma_range(ic<CUR>{}, ic<0>{}, ic<QH>{});
        __builtin_amdgcn_sched_barrier(0);
        if (!(xp & 4)) load_frags(fa[0], fb[0], smem + ((kt + 1) & 1) * STAGE, 0);    // (stale LDS after the last K-tile: unused)
        pending = kt + 2 < nk && !(xp & 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NP > 0) {
        // the pending K-tile is kt+2 at k-step 3, and (this tile)+1 at k-steps 0/1 of the following tile; its buffer has the
        // parity of the tile that was current when the hand-over released it
        const int ikt = ks == 3 ? kt + 2 : kt + 1;
        const int ibuf = ikt & 1;
        static_for<NP>([&](auto d_c) {
          constexpr int d = decltype(d_c)::value;
          mfma_range(ic<CUR>{}, ic<(QH + d < Q ? QH + d : Q)>{}, ic<(QH + 1 + d < Q ? QH + 1 + d : Q)>{});
          if (pending) issue_piece(ibuf, ikt, ic<FP + d>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        mfma_range(ic<CUR>{}, ic<(QH + NP < Q ? QH + NP : Q)>{}, ic<Q>{});
        if constexpr (FP + NP == P) pending = false;
      } else {
        mfma_range(ic<CUR>{}, ic<QH>{}, ic<Q>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
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
  dim3 grid((unsigned)(tiles_m * tiles_n * zs)), block(WM * WN * 64);
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
  if (k == 13) { launch_pipe<256, 256, 2, 4, MODE, 1>(g, ep, s); return; }
  if (k == 14) { launch_pipe<256, 256, 2, 4, MODE, 2>(g, ep, s); return; }
  if (k == 15) { launch_pipe<320, 256, 2, 4, MODE, 2>(g, ep, s); return; }
  if (k == 9) { launch_pipe<256, 256, 2, 4, MODE>(g, ep, s); return; }
  if (k == 10) { launch_pipe<320, 256, 2, 4, MODE>(g, ep, s); return; }
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
  if (g_shared_gpu) return (g_allow_320 && e320 > e256 * 1.08) ? 5 : 2;   // the hardware dispatcher balances one-tile workgroups
  return (g_allow_320 && e320 > e256 * 1.08) ? 7 : 6;                      // persistent variants
}
int gemm_bf16_tile_m(int kernel, int M, int N) {
  kernel &= 15;
  if (kernel == 0) kernel = gemm_bf16_pick(M, N);
  return kernel == 1 ? 128 : ((kernel == 5 || kernel == 7 || kernel == 10 || kernel == 15) ? 320 : 256);
}
int gemm_bf16_tile_n(int kernel, int M, int N) {
  kernel &= 15;
  if (kernel == 0) kernel = gemm_bf16_pick(M, N);
  return (kernel == 1 || kernel == 3) ? 128 : 256;
}
void gemm_bf16_allow_320(int on) { g_allow_320 = on; }
void gemm_bf16_set_shared_gpu(int on) { g_shared_gpu = on; }

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

// ---- per-shape variant selection by measurement.  The variants differ by a few percent per (shape, epilogue) and the ranking
// moves with the board's clocks, so with kernel = 0 (automatic) the first launch of each (mode, M, N, K) times the candidates on
// the caller's stream with the caller's operands (every fused epilogue is a pure function of its inputs -- the launch is
// repeatable as long as the output does not alias the residual) and caches the winner for the process.  All variants accumulate
// every output element in the same K order, so the choice does not change results.  VITX_GEMM_AUTOTUNE=0 falls back to the
// static rule gemm_bf16_pick().
static std::mutex g_tune_mu;
static std::map<std::array<int64_t, 6>, int> g_tuned;
static int autotune_enabled() {
  static int on = -1;
  if (on < 0) { const char* v = getenv("VITX_GEMM_AUTOTUNE"); on = (v && atoi(v) == 0) ? 0 : 1; }
  return on;
}
void launch_gemm_bf16(const Bf16GemmArgs& g0, const EpiParams& ep, int mode, hipStream_t s) {
  // small batches: a few hundred token rows give the 256x256 tiles less than two per CU, where the smaller tiles can win by a lot
  const bool few_tiles = ceil_div(g0.M, 256) * ceil_div(g0.N, 256) < 512;
  const double work = (double)g0.M * g0.N * g0.K;
  const bool tunable = g0.kernel == 0 && autotune_enabled() && (g0.N % 256 == 0 || g0.N > 512) && (work >= 2.0e9 || (few_tiles && work >= 1.0e8)) &&
                       !(mode == EPI_BIAS_RESID && ep.out == (void*)ep.resid);
  if (!tunable) { dispatch_gemm_bf16(g0, ep, mode, s); return; }
  const std::array<int64_t, 6> key = {mode, g0.M, g0.N, g0.K, g0.split_k > 1 ? g0.split_k : 1, (ep.scale != nullptr) + 2 * g_shared_gpu};
  int best = -1;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) best = it->second;
  }
  Bf16GemmArgs g = g0;
  if (best < 0) {
    static const int cand[] = {6, 13, 14, 2, 7, 15, 10, 5, 3, 1};   // 256x256 variants first, then 320x256 (only when allowed), then small tiles
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
    for (int c : cand) {
      const bool is320 = c == 5 || c == 7 || c == 10 || c == 15;
      if (is320 && !g_allow_320) continue;
      if ((c == 1 || c == 3) && !small_m) continue;
      if (g_shared_gpu && c != 2 && c != 5 && c != 1 && c != 3) continue;   // no persistent variants beside collectives (see gemm_bf16_set_shared_gpu)
      g.kernel = c;
      dispatch_gemm_bf16(g, ep, mode, s);   // warm-up (first-use attribute setup, instruction cache)
      (void)hipEventRecord(ev[0], s);
      for (int r = 0; r < NREP; ++r) {
        dispatch_gemm_bf16(g, ep, mode, s);
        (void)hipEventRecord(ev[r + 1], s);
      }
      if (hipEventSynchronize(ev[NREP]) != hipSuccess) continue;
      float fastest = 1e30f;
      for (int r = 0; r < NREP; ++r) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[r], ev[r + 1]) == hipSuccess && ms < fastest) fastest = ms;
      }
      if (fastest < best_ms) { best_ms = fastest; best = c; }
    }
    for (auto& x : ev) (void)hipEventDestroy(x);
    if (getenv("VITX_GEMM_AUTOTUNE_LOG"))
      fprintf(stderr, "[vitx] gemm autotune: mode %d M %d N %d K %d split %d -> variant %d (%.4f ms)\n", mode, g0.M, g0.N, g0.K, g0.split_k, best,
              best_ms);
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = best;
  }
  g.kernel = best;
  dispatch_gemm_bf16(g, ep, mode, s);
}

// C[M=in][N=out] (split-K partials) = A[K=tokens][in]^T * B[K=tokens][out]; kernel: 1 = 128x128 tile, else 256x256
int gemm_bf16_tn_tile(int kernel, int M, int N) { kernel &= 15; return (kernel == 1 || (M <= 128 && N <= 128)) ? 128 : 256; }
void launch_gemm_bf16_tn(const Bf16GemmArgs& g0, const EpiParams& ep, hipStream_t s) {
  static const int xp = [] {
    const char* v = getenv("VITX_TN_XP");
    const int x = v ? atoi(v) : 0;
    if (x) fprintf(stderr, "[vitx] VITX_TN_XP=%d: timing experiment -- weight gradients are WRONG in this process\n", x);
    return x;
  }();
  Bf16GemmArgs g = g0;
  g.stagger = xp;
  if (gemm_bf16_tn_tile(g.kernel, g.M, g.N) == 128) launch_tn_variant<128, 2, 2>(g, ep, s);
  else launch_tn_variant<256, 2, 4>(g, ep, s);
}
