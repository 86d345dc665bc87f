// Wave-group "ping-pong" NT kernel of the bf16 MFMA GEMM family (variant 12 of launch_gemm_bf16; design notes of the family: gemm_bf16.hip).
//
// Why: in the persistent 8-wave kernels (gemm_bf16_pipe.hip) the epilogue of a tile runs in series with the K loops, with the matrix pipe
// idle and the per-CU store path (~16 B/clk) as the only thing working -- 25-45 % of the K = 768 launches of a ViT block -- and the MFMA
// waves issue the LDS-DMA themselves (1284 -> 1570 TFLOP/s at 8192^3 when they do not).  Here the eight waves of a workgroup are two groups
// of four (one wave per SIMD each) that take turns:
//   * CONSUMER group: K loop of ITS 256 x 128 output tile (wave tile 128 x 64): MFMAs and fragment ds_reads, no vector-memory instruction at all;
//   * PRODUCER group: issues every LDS-DMA piece of the consumer's K-tile stream, and meanwhile drains the accumulators of the tile it
//     computed in the previous half-step (one 8-row group per phase: LDS transpose in a private 8-KiB area, epilogue arithmetic, <= 2 global
//     stores per wave and phase so that the store path never makes it late for the barrier).
// After nk phases (one per K-tile of 32) the roles swap.  The matrix pipe always has one wave per SIMD feeding it and the epilogue's HBM time
// hides under the other group's K loop.  Price: a 256 x 128 tile per K-tile stream = 1.5x the DMA bytes per FLOP of a 256 x 256 tile.
//
// Stream / synchronisation (g = running K-tile index of the workgroup, slot = g % 4, BK = 32 so a slot is (256 + 128) rows x 64 B = 24 KiB):
//   consumer, K-tile g:   k-step 0: 8 MFMAs (fragments F0(g)), reads F1(g) behind the 2nd
//                         k-step 1: lgkmcnt(0); 2 MFMAs; BARRIER(g); reads F0(g+1); 6 MFMAs
//   producer, phase g:    issue K-tile g+3 into slot (g-1)%4 (released by BARRIER(g-1): every read of K-tile g-1 retired before it);
//                         one epilogue micro-step; wait until K-tile g+1 has landed (counted vmcnt: memory operations of a wave retire in
//                         order, so "all but the N youngest" is exact); BARRIER(g)
//   The wave that ISSUED a K-tile waits for it: across a role swap the first two K-tiles of a half-step were issued by the group that is now
//   the consumer, which therefore waits for them (vmcnt(6), vmcnt(0) -- it has nothing else in flight) before its first two barriers.
// All DMA goes through vitx_dma16 (inline asm, invisible to the compiler's waitcnt pass); the waits are the modelled builtin.
//
// RESULT (MI355X, round 3): correct on every shape (tests/test_gpu_full_size.py, tools/pp_check.py) but SLOWER than the 256 x 256 kernels:
// 928 vs 1268-1306 TFLOP/s at 8192^3, fc1 + GELU 431 vs 376 us.  The consumer loop alone runs at 1447-1496 TFLOP/s (timing switch 2); what
// binds is the operand feed: 24 KiB per 16 MFMAs per SIMD = ~28 B/clk/CU at 1300 TFLOP/s, while a CU gets 35 (4 issuing waves) - 49 B/clk
// through the LDS-DMA from an L2-RESIDENT source with full-line pieces, ~30 with the half-line pieces of BK = 32, and only 14 B/clk (7.5 TB/s
// for the whole chip) for whatever misses L2 (tools/probe_feed.hip, profiles/r3/probe_feed_paths_r3j.log).  A variant that staged the operands
// through the producer's registers (buffer_load -> ds_write_b128) was slower still (737).  The structure hides the epilogue but pays 1.5x the
// operand bytes per FLOP for it, which this memory system does not have to give.  Kept as a tested variant (12), not an autotune candidate.
//
// Eligible (launch_gemm_bf16_pp returns false otherwise and the caller takes another variant): interior-only problems (M % 256 == 0,
// N % 128 == 0), K % 32 == 0 with at least PP_MIN_NK K-tiles, 16-B aligned rows, alpha == 1, no LayerScale, no split-K, operands < 2 GiB;
// epilogues EPI_STORE (no bias), EPI_BIAS_GELU (bf16 form: act + stored gelu'), EPI_BIAS_RESID, EPI_GELU_BWD (bf16 form, fused column sums).
#include "gemm_bf16_common.h"

namespace {

constexpr int PBM = 256, PBN = 128, PBK = 32;
constexpr int PNS = 4;                                   // ring slots
constexpr int PSLOT = (PBM + PBN) * PBK * 2;             // 24576 B: A rows at +0 (16 KiB), B rows at +16384 (8 KiB)
constexpr int PRING = PNS * PSLOT;                       // 98304 B
constexpr int PSTAGE = 8192;                             // private epilogue staging per wave: 32 rows x 64 fp32
constexpr int PSMEM = PRING + 8 * PSTAGE;                // 163840 B = all of the CU's LDS
constexpr int PP_U = 16;                                 // epilogue micro-steps per tile (4 block rows x 4 groups of 8 rows)
constexpr int PP_MIN_NK = PP_U + 3;                      // micro-steps in phases 0 .. 15, then >= 2 store-free phases + the zeroing phase before the role swap

constexpr int waitcnt_vm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n) only
constexpr int WAIT_LGKM0 = 0xC07F;

template <int MODE>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(Bf16GemmArgs g, EpiParams ep, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int pair0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // this workgroup's first tile pair
  const int total_tiles = tiles_m * tiles_n;
  const int nk = g.K / PBK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
  // Per-lane address pieces are derived from an OPAQUE copy of the lane id inside each role / micro-step: derived from `lane` itself they are
  // loop invariants that the compiler hoists to the top of the kernel and keeps live through both roles (~60 VGPRs beside 128 accumulators and
  // 48 fragment registers: 40-90 spilled VGPRs, whose scratch traffic would also sit in the vmcnt queue the producer's waits count on).
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  const int gm = (tiles_n >= 8 ? 4 : 1) * (g.reverse_m ? -1 : 1);

  // tile of half-step h: pair = pair0 + (h >> 1) * nwg, tile = 2 * pair + (h & 1); the sequence ends at the first tile that does not exist
  auto tile_of = [&](int h) { return 2 * (pair0 + (h >> 1) * nwg) + (h & 1); };

  // ---------------------------------------------------------------- DMA (producer side)
  const i32x4 rsA = vitx_make_rsrc(g.A), rsB = vitx_make_rsrc(g.B);
  const uint32_t lds0 = vitx_lds_addr(smem);
  // piece p (1 KiB) of an operand covers rows 16p .. 16p+15 of the slot image: lane -> (row 16p + lane/4, physical 16-B chunk lane%4), which
  // holds logical chunk  pc ^ ((row >> 2) & 3)  (bank-conflict swizzle of the 64-B rows, applied to the SOURCE address and to the reads)
  uint32_t offA[4], offB[2];   // wave w4 issues A pieces w4, w4+4, w4+8, w4+12 and B pieces w4, w4+4
  auto set_dma_offsets = [&]() {
    const int l = opaque(lane);
    const int prow = l >> 2, pchunk = (l & 3) ^ ((l >> 4) & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) offA[j] = (uint32_t)((16 * (w4 + 4 * j) + prow) * (int)g.lda + pchunk * 8) * 2u;
#pragma unroll
    for (int j = 0; j < 2; ++j) offB[j] = (uint32_t)((16 * (w4 + 4 * j) + prow) * (int)g.ldb + pchunk * 8) * 2u;
  };
  // The stream has no state: K-tile number n of the workgroup (n = half_step * nk + k) belongs to the tile of half-step n / nk, and in phase
  // (h, k) the producer issues K-tile h * nk + k + 3 -- of tile h, or of tile h + 1 for the last three phases (nk >= 3).
  int n_tiles = 0;
  while (tile_of(n_tiles) < total_tiles) ++n_tiles;
  const int total_items = n_tiles * nk;
  auto tile_offsets = [&](int h, uint32_t& a_soff, uint32_t& b_soff) {
    int tm, tn;
    decode_tile(tile_of(h), tiles_m, tiles_n, gm, tm, tn);
    a_soff = (uint32_t)((int64_t)tm * PBM * g.lda * 2);
    b_soff = (uint32_t)((int64_t)tn * PBN * g.ldb * 2);
  };
  const int xp = g.stagger;   // timing experiments only (results are wrong): 1 = the producer never waits for its pieces, 2 = no pieces in the K loop
  auto issue_item = [&](int n, uint32_t a_soff, uint32_t b_soff, int ik) {   // K-tile n -> slot n % 4
    if ((xp & 2) && n >= 3) return;
    const uint32_t base = lds0 + (uint32_t)(n & (PNS - 1)) * PSLOT;
    const uint32_t ko = (uint32_t)ik * (PBK * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) vitx_dma16(rsA, base + (uint32_t)(w4 + 4 * j) * 1024u, offA[j], a_soff + ko);
#pragma unroll
    for (int j = 0; j < 2; ++j) vitx_dma16(rsB, base + 16384u + (uint32_t)(w4 + 4 * j) * 1024u, offB[j], b_soff + ko);
  };

  // ---------------------------------------------------------------- fragments (consumer side)
  int fsw = 0, a_row_byte = 0, b_row_byte = 0, fkh = 0;   // set per consumer role (set_frag_offsets)
  auto set_frag_offsets = [&]() {
    const int l = opaque(lane);
    const int m32 = l & 31;
    fkh = l >> 5;
    fsw = (m32 >> 2) & 3;
    a_row_byte = (wm * 128 + m32) * 64;
    b_row_byte = 16384 + (wn * 64 + m32) * 64;
  };
  bf16x8 fa[2][4], fb[2][2];
  auto load_frags = [&](bf16x8(&af)[4], bf16x8(&bfr)[2], const char* base, int ks) {
    const int cb = ((ks * 2 + fkh) ^ fsw) << 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(base + a_row_byte + i * 32 * 64 + cb);
#pragma unroll
    for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(base + b_row_byte + j * 32 * 64 + cb);
  };
  // The accumulators are never "zeroed": the first k-step of a tile runs its MFMAs with C = 0 (ZERO_C), so they are defined by the consumer
  // role and dead after the drain -- zeroing them in the producer role made them phi nodes of two register sets (copies + spills).
  f32x16 acc[4][2];
  auto mfma_range = [&](auto cur_c, auto first_c, auto last_c, auto zero_c) {
    constexpr int CUR = decltype(cur_c)::value, FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
    constexpr bool ZERO_C = decltype(zero_c)::value != 0;
    static_for<LAST - FIRST>([&](auto d) {
      constexpr int idx = FIRST + decltype(d)::value, i = idx / 2, j = idx % 2;
      if constexpr (ZERO_C) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][j], fa[CUR][i], z, 0, 0, 0);
      } else {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[CUR][j], fa[CUR][i], acc[i][j], 0, 0, 0);
      }
    });
  };

  // ---------------------------------------------------------------- epilogue state (tile computed in the previous half-step)
  int my_tm = 0, my_tn = 0;                  // tile this group consumed last
  float4 bias_lo = make_float4(0.f, 0.f, 0.f, 0.f), bias_hi = bias_lo;   // bias of this lane's 8 output columns (row-major domain)
  char* const stg = smem + PRING + wave * PSTAGE;
  const bool has_bias = ep.bias != nullptr;

  // micro-step u = 4 i + t: (t == 0: accumulators of block row i -> staging, fp32, 16-B chunks swizzled chunk ^= row & 15), then the 8 rows 8t .. 8t+7 of the block come back row-major (8 consecutive columns per lane) and go through the epilogue
  auto micro_step = [&](auto u_c) {
    constexpr int u = decltype(u_c)::value, i = u >> 2, t = u & 3;
    const int l = opaque(lane);
    const int m32 = l & 31, khalf = l >> 5;
    const int rr = l >> 3, cp = l & 7;   // row-major read-back: row 8t + rr of the 32-row block, logical columns 8 cp .. 8 cp + 7 of the wave's 64
    if constexpr (t == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = j * 8 + 2 * q + khalf;
          *(float4*)(stg + m32 * 256 + ((c ^ (m32 & 15)) << 4)) =
              make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        }
    }
    const int r = 8 * t + rr;
    float4 lo = *(const float4*)(stg + r * 256 + (((2 * cp) ^ (r & 15)) << 4));
    float4 hi = *(const float4*)(stg + r * 256 + (((2 * cp + 1) ^ (r & 15)) << 4));
    const int64_t grow = (int64_t)my_tm * PBM + wm * 128 + i * 32 + r;
    const int gcol = my_tn * PBN + wn * 64 + cp * 8;
    if (MODE == EPI_BIAS_GELU || MODE == EPI_BIAS_RESID) {
      lo.x += bias_lo.x; lo.y += bias_lo.y; lo.z += bias_lo.z; lo.w += bias_lo.w;
      hi.x += bias_hi.x; hi.y += bias_hi.y; hi.z += bias_hi.z; hi.w += bias_hi.w;
    }
    if constexpr (MODE == EPI_STORE) {
      bf16x8* o = (bf16x8*)((bf16_t*)ep.out + grow * ep.ldo + gcol);
      if (ep.nt_out) __builtin_nontemporal_store(pack_bf16x8(lo, hi), o);
      else *o = pack_bf16x8(lo, hi);
    } else if constexpr (MODE == EPI_BIAS_GELU) {
      const bf16x8 h = pack_bf16x8(lo, hi);       // GELU and its derivative of the pre-activation as bf16 would store it (as every other variant)
      float4 g0, d0, g1, d1;
      gelu_both4(make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]), g0, d0);
      gelu_both4(make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]), g1, d1);
      bf16x8* od = (bf16x8*)((bf16_t*)ep.out + grow * ep.ldo + gcol);
      if (ep.nt_out) __builtin_nontemporal_store(pack_bf16x8(d0, d1), od);
      else *od = pack_bf16x8(d0, d1);
      *(bf16x8*)((bf16_t*)ep.out2 + grow * ep.ldo2 + gcol) = pack_bf16x8(g0, g1);
    }
  };
  constexpr int ST = (MODE == EPI_BIAS_GELU) ? 2 : 1;   // global stores of one micro-step (the exact count the producer's vmcnt arithmetic needs)

  // ---------------------------------------------------------------- prologue: the producer of half-step 0 (group 1) fills three slots
  if (n_tiles == 0) return;
  if (grp == 1) {
    uint32_t a0, b0;
    set_dma_offsets();
    tile_offsets(0, a0, b0);
    issue_item(0, a0, b0, 0);
    issue_item(1, a0, b0, 1);
    issue_item(2, a0, b0, 2);
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(12));   // K-tile 0 landed
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int gbase = 0;   // running K-tile index of the first K-tile of the current half-step
  for (int h = 0;; ++h, gbase += nk) {
    const bool have_tile = tile_of(h) < total_tiles;
    if (!have_tile) {
      // last half-step: the group that computed the last tile drains it (no stream, no barriers); the other group is done
      if (h > 0 && grp == ((h - 1) & 1)) static_for<PP_U>([&](auto u_c) { micro_step(u_c); });
      break;
    }
    if (grp == (h & 1)) {
      // ================================================================ consumer
      decode_tile(tile_of(h), tiles_m, tiles_n, gm, my_tm, my_tn);
      int slot = gbase & (PNS - 1);
      set_frag_offsets();
      load_frags(fa[0], fb[0], smem + slot * PSLOT, 0);
      auto item = [&](int k, auto first_c) {
        constexpr int FIRST = decltype(first_c)::value;        // 1: first K-tile of the tile (k == 0): its first k-step starts the accumulators
        const char* base = smem + slot * PSLOT;
        // ---- k-step 0
        mfma_range(ic<0>{}, ic<0>{}, ic<2>{}, ic<FIRST>{});
        __builtin_amdgcn_sched_barrier(0);
        load_frags(fa[1], fb[1], base, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(ic<0>{}, ic<2>{}, ic<8>{}, ic<FIRST>{});
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-step 1: every read of this K-tile retired, then (first two phases after a role swap) the K-tiles this group issued as producer:
        // K-tile gbase+1 (issued two phases ago, needed behind this barrier) and gbase+2 (one phase ago); its stores ended >= 2 phases before the swap
        __builtin_amdgcn_s_waitcnt(WAIT_LGKM0);
        if constexpr (FIRST) { if (h > 0) __builtin_amdgcn_s_waitcnt(waitcnt_vm(6)); }
        else { if (h > 0 && k == 1) __builtin_amdgcn_s_waitcnt(waitcnt_vm(0)); }
        if constexpr (!FIRST && (MODE == EPI_BIAS_GELU || MODE == EPI_BIAS_RESID)) {
          if (k == 2 && has_bias) {
            // bias of this lane's 8 row-major columns: an ordinary load, issued when nothing else of this wave is in flight and waited for at the
            // end of the K loop (the consumer issues no other vector-memory instruction, so the wait cannot catch a DMA piece)
            const float* bp = ep.bias + my_tn * PBN + wn * 64 + (opaque(lane) & 7) * 8;
            bias_lo = *(const float4*)bp;
            bias_hi = *(const float4*)(bp + 4);
          }
        }
        mfma_range(ic<1>{}, ic<0>{}, ic<2>{}, ic<0>{});
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        slot = (slot + 1) & (PNS - 1);
        if (k + 1 < nk) load_frags(fa[0], fb[0], smem + slot * PSLOT, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(ic<1>{}, ic<2>{}, ic<8>{}, ic<0>{});
        __builtin_amdgcn_sched_barrier(0);
      };
      item(0, ic<1>{});
      for (int k = 1; k < nk; ++k) item(k, ic<0>{});
      __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));   // (bias load; modelled, so the compiler enters the producer role with an empty scoreboard)
    } else {
      // ================================================================ producer
      uint32_t a_cur, b_cur, a_nxt = 0, b_nxt = 0;
      set_dma_offsets();
      tile_offsets(h, a_cur, b_cur);
      if (h + 1 < n_tiles) tile_offsets(h + 1, a_nxt, b_nxt);
      auto issue_phase = [&](int k) {              // K-tile gbase + k + 3, into the slot released by the barrier this wave has just passed
        const int n = gbase + k + 3;
        if (n < total_items) {
          if (k + 3 < nk) issue_item(n, a_cur, b_cur, k + 3);
          else issue_item(n, a_nxt, b_nxt, k + 3 - nk);
        }
      };
      // K-tile gbase + k + 1 must have landed before the barrier of phase k.  This wave issued it iff that was >= 2 phases into this half-step
      // (or in the prologue of half-step 0); younger than it in this wave's queue: the two K-tiles issued since (6 pieces each) and `st` stores.
      auto wait_phase = [&](int k, auto st_c) {
        constexpr int st = decltype(st_c)::value;
        if (xp & 1) return;
        if (total_items - (gbase + k + 2) >= 2) __builtin_amdgcn_s_waitcnt(waitcnt_vm(12 + st));
        else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));             // the last two K-tiles of the whole stream
      };
      auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      };
      if (h == 0) {
        for (int k = 0; k < nk; ++k) {             // nothing to drain yet; this group issued the prologue's K-tiles itself
          issue_phase(k);
          wait_phase(k, ic<0>{});
          barrier();
        }
      } else {
        // phases 0 .. 15: one epilogue micro-step each, with COMPILE-TIME indices (as cases of a run-time switch the sixteen near-identical
        // bodies were merged by the optimiser into one block fed by phi nodes over the accumulators: 300-500 spilled VGPRs)
        static_for<PP_U>([&](auto k_c) {
          constexpr int k = decltype(k_c)::value;
          issue_phase(k);
          micro_step(k_c);
          if constexpr (k >= 2) wait_phase(k, ic<3 * ST>{});
          barrier();
        });
        for (int k = PP_U; k < nk - 1; ++k) {
          issue_phase(k);
          if (k == PP_U) wait_phase(k, ic<2 * ST>{});
          else if (k == PP_U + 1) wait_phase(k, ic<ST>{});
          else wait_phase(k, ic<0>{});
          barrier();
        }
        issue_phase(nk - 1);
        wait_phase(nk - 1, ic<0>{});
        barrier();
      }
    }
  }
}

template <int MODE>
void launch_pp_mode(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  auto kern = gemm_bf16_pp_kernel<MODE>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PSMEM);
    attr_set = true;
  }
  const int tiles_m = g.M / PBM, tiles_n = g.N / PBN;
  const int pairs = (tiles_m * tiles_n + 1) / 2;
  static const int grid_cap = [] { const char* v = getenv("VITX_GEMM_GRID"); return v ? atoi(v) : 256; }();
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min(pairs, grid_cap)), dim3(512), PSMEM, s, g, ep, tiles_m, tiles_n);
}

}  // namespace

bool gemm_bf16_pp_eligible(const Bf16GemmArgs& g, const EpiParams& ep, int mode) {
  if (mode != EPI_STORE && mode != EPI_BIAS_GELU) return false;
  if (mode == EPI_STORE && ep.bias != nullptr) return false;
  if (g.M % PBM || g.N % PBN || g.K % PBK || g.K / PBK < PP_MIN_NK || g.split_k > 1) return false;
  if (ep.M != g.M || ep.N != g.N || !ep.vec_ok || !ep.wide_ok || ep.alpha != 1.0f || ep.scale != nullptr) return false;
  if (((int64_t)g.M * g.lda + g.K) * 2 >= (1LL << 31) || ((int64_t)g.N * g.ldb + g.K) * 2 >= (1LL << 31)) return false;
  return true;
}

bool launch_gemm_bf16_pp(const Bf16GemmArgs& g, const EpiParams& ep, int mode, hipStream_t s) {
  if (!gemm_bf16_pp_eligible(g, ep, mode)) return false;
  if (mode == EPI_STORE) launch_pp_mode<EPI_STORE>(g, ep, s);
  else launch_pp_mode<EPI_BIAS_GELU>(g, ep, s);
  return true;
}
