// Software-pipelined persistent NT kernel of the bf16 MFMA GEMM family (see gemm_bf16.hip for the family's design notes).
#include "gemm_bf16_common.h"

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

// lane index from the hardware, opaque to the optimiser (no live range across the K loop, nothing derived from it is loop-invariant)
__device__ __forceinline__ int wp_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// 16-byte output store by cache policy: 0 plain, 1 non-temporal.  (Write-through `sc1` stores -- the line is not kept in the XCD's L2 -- were measured in
// round 5 as a way to keep operand panels L2-resident across a workgroup's tiles: FETCH_SIZE -5 %, time unchanged, 20-40 more SGPR spills in the GELU
// instantiation; not kept, profiles/r5/ab_write_through_stores_*_r5c_not_kept.log.)
template <int POL>
__device__ __forceinline__ void wp_store16(bf16_t* base, uint32_t elem_off, bf16x8 v) {
  if constexpr (POL == 1) __builtin_nontemporal_store(v, (bf16x8*)(base + elem_off));
  else *(bf16x8*)(base + elem_off) = v;
}

// ---- GELU by table (the fc1 epilogue, 256-row tiles).  The pre-activation is rounded to bf16 BEFORE the activation is applied (what a bf16
// tensor of it would hold), so gelu / gelu' are functions of a 16-bit pattern: for 2^-12 <= |x| < 8 (15 binades x 128 mantissas x 2 signs = 3840
// patterns) they are read from a 15-KiB LDS table; smaller |x| take the first entry of their sign (Phi = 1/2, gelu' = 1/2 to bf16 precision), larger
// ones the last (Phi = 1 or 0, gelu' = 1 or 0).  Entry = Phi(x) as fp16 (high half) | gelu'(x) as bf16 (low half); gelu(x) = x * Phi(x), so both
// ends are exact without a fix-up.  Built on the host from erf in double precision (closer to the exact-erf GELU of vit.py:34 than the degree-7
// polynomial of the other epilogues: |Phi error| <= 2.4e-4 relative, below a quarter of a bf16 ulp).  Why: the polynomial form costs 17.5 VALU issue
// slots per element (two v_exp_f32 per pair at quarter rate), the epilogue of a 256 x 256 tile 18k cycles per SIMD next to a 33k-cycle K loop; the
// table form ~11 slots and one LDS gather per element (the LDS is idle in the epilogue).
constexpr int GT_LO = (127 - 12) << 7, GT_NE = ((127 + 3) << 7) - GT_LO, GT_BYTES = 2 * GT_NE * 4;   // 14720, 1920 entries per sign, 15360 B
const uint32_t* gelu_table_dev() {   // one copy per device (one process may drive several handles on different GPUs)
  static std::mutex mu;
  static std::map<int, const uint32_t*> per_dev;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  auto it = per_dev.find(dev);
  if (it != per_dev.end()) return it->second;
  std::vector<uint32_t> h((size_t)2 * GT_NE);
  auto f32_to_bf16 = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
  auto f32_to_f16 = [](float f) { _Float16 hh = (_Float16)f; uint16_t u; memcpy(&u, &hh, 2); return u; };
  for (int sgn = 0; sgn < 2; ++sgn)
    for (int t = 0; t < GT_NE; ++t) {
      const uint32_t bits = ((uint32_t)sgn << 31) | ((uint32_t)(t + GT_LO) << 16);
      float xf; memcpy(&xf, &bits, 4);
      const double x = xf, phi = 0.5 * (1.0 + erf(x * 0.70710678118654752440)), gd = phi + x * 0.39894228040143267794 * exp(-0.5 * x * x);
      h[(size_t)sgn * GT_NE + t] = ((uint32_t)f32_to_f16((float)phi) << 16) | f32_to_bf16((float)gd);
    }
  uint32_t* d = nullptr;
  if (hipMalloc((void**)&d, GT_BYTES) != hipSuccess || hipMemcpy(d, h.data(), GT_BYTES, hipMemcpyHostToDevice) != hipSuccess) d = nullptr;   // null: the polynomial form
  per_dev[dev] = d;
  return d;
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
//   * the epilogue is wave-private (no workgroup barrier): see the epilogue section.
template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_bf16_nt_pipe_kernel(Bf16GemmArgs g, EpiParams ep, int tiles_m, int tiles_n,
                                                                         int kt_per_split, const uint32_t* __restrict__ gelu_tab) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
  static_assert(BN == 256 || BN == 128, "256-column tiles (8 waves) or 128-column tiles (4 waves, two workgroups per CU)");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int logical0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int total_tiles = tiles_m * tiles_n;
  // ---- the workgroup's tile list: indices w_first, w_first + w_stride, ... < w_limit of a walk.
  // walk 0: the XCD-interleaved chunks of the global list (decode_tile: row-major, or bands of 4 row tiles for wide outputs).
  // walk 2 (wide outputs, full persistent grid; the launcher decides): every XCD OWNS a contiguous range of row tiles and its workgroups walk it in
  // bands of 8 row tiles, column-major inside a band -- the 32 tiles an XCD works on at any time are 8 rows x 4 columns of ONE band, and the band's
  // 8 A panels (3 MiB at K = 768) stay in that XCD's L2 across the band's rounds instead of every round fetching 4 + 8 fresh panels
  // (PMC, round 5: fc1 / qkv / fc2-dgrad fetched their A operand 4-5 times through the L2s; profiles/r5/fetch_by_shape_*.txt).
  const bool xw = BM == 256 && g.walk == 2;   // (256-row tiles only: the 320-row instantiations serve 768-wide outputs and have no register to spare)
  const int w_r0 = xw ? (tiles_m * xcd) >> 3 : 0, w_nr = xw ? ((tiles_m * (xcd + 1)) >> 3) - w_r0 : tiles_m;
  const int w_first = xw ? (bid >> 3) : logical0, w_stride = xw ? (nwg >> 3) : nwg, w_limit = xw ? w_nr * tiles_n : total_tiles;
  if (w_first >= w_limit) return;
  const int z = blockIdx.y;
  const int kt0 = z * kt_per_split;
  const int nk = min(kt_per_split, g.K / BK - kt0);
  const bool persistent = gridDim.y == 1;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;

  if (g.phase > 0 && persistent) {   // de-phase the workgroups of an XCD: their store-heavy epilogues stop coinciding
    // phase < 100: eight phases, ((bid >> 3) & 7) * phase * 4096 cycles; phase >= 100: two phases, every other CU slot of an XCD starts
    // (phase - 100) * 4096 cycles late (about half a tile period: half the chip is in its K loop while the other half stores)
    const int n = g.phase >= 100 ? ((bid >> 3) & 1) * (g.phase - 100) : ((bid >> 3) & 7) * g.phase;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(64);
  }

  // per-lane DMA source offsets in BYTES, unsigned: address = uniform 64-bit base (SGPR pair) + zero-extended 32-bit lane offset,
  // which selects the saddr form of global_load_lds (one address dword per lane, no per-piece VALU address arithmetic)
  uint32_t offA[A_INSTR], offB[B_INSTR];
#pragma unroll
  // wave w moves rows [8 w A_INSTR, 8 (w + 1) A_INSTR) of the A tile: its pieces are back to back in LDS, so that groups of four share ONE M0
  // value and differ in the instruction's immediate offset (vitx_dma16_cont; the offset also shifts the global address, hence the - (i & 3) KiB)
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (wave * A_INSTR + i) * 8 + (lane >> 3);
    offA[i] = (uint32_t)(row * (int)g.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2u - (uint32_t)((i & 3) * 1024);
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int row = (wave * B_INSTR + i) * 8 + (lane >> 3);
    offB[i] = (uint32_t)(row * (int)g.ldb + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2u - (uint32_t)((i & 3) * 1024);
  }

  const int gm = ((tiles_n >= 8 && g.stagger != 8) ? 4 : 1) * (g.reverse_m ? -1 : 1);   // (stagger 8: row-major order, A/B switch of the micro-benchmark)
  // ---- issue cursor over this workgroup's K-tile stream (tile, kt); LDS buffer of stream item i = i & 1
  auto tile_of = [&](int idx, int& tm, int& tn) {
    if (xw) {
      decode_tile_fwd(idx, w_nr, tiles_n, 8, tm, tn);
      tm += w_r0;
      if (g.reverse_m) tm = tiles_m - 1 - tm;
    } else {
      decode_tile(idx, tiles_m, tiles_n, gm, tm, tn);
    }
  };
  int i_logical = w_first, i_k = 0, issued = 0;
  bool i_more = nk > 0;
  // DMA addressing: buffer_load ... lds with one resource per operand (SGPRs), the tile / K offset in the scalar offset and a
  // loop-invariant 32-bit lane offset -- no per-piece VALU address arithmetic and one address dword per lane instead of two
  // (the flat global_load_lds form needs a 64-bit address per lane).  Operands are < 4 GiB (checked by the launcher).
  // The pieces are issued from inline asm (vitx_dma16, common.h): as builtins the compiler drained them with vmcnt(0) in front of the
  // next k-step's fragment reads.
  const i32x4 rsA = vitx_make_rsrc(g.A), rsB = vitx_make_rsrc(g.B);
  const uint32_t lds_wa = vitx_lds_addr(smem) + (uint32_t)wave * (A_INSTR * 1024u);             // this wave's A pieces (contiguous)
  const uint32_t lds_wb = vitx_lds_addr(smem) + A_BYTES + (uint32_t)wave * (B_INSTR * 1024u);   // this wave's B pieces
  uint32_t a_soff = 0, b_soff = 0;       // byte offset of the cursor tile's first K-tile inside A / B
  auto i_set_tile = [&]() {
    int tm, tn;
    tile_of(i_logical, tm, tn);
    a_soff = (uint32_t)(((int64_t)tm * BM * g.lda + (int64_t)kt0 * BK) * 2);
    b_soff = (uint32_t)(((int64_t)tn * BN * g.ldb + (int64_t)kt0 * BK) * 2);
    {
      // The uniformity analysis loses the cursor behind the lane-dependent branches of the wave-private epilogue and would hand the asm a VGPR
      // for the scalar offset.  Made scalar HERE, once per tile: a v_readfirstlane right in front of a piece is a VALU write of an SGPR that the
      // buffer_load inside the asm reads as its scalar offset -- a 5-wait-state hazard the compiler cannot see (pieces fetched from stale offsets).
      a_soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)a_soff);
      b_soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_soff);
    }
  };
  i_set_tile();
  constexpr int P = A_INSTR + B_INSTR;   // DMA pieces (1 KiB each) per K-tile per wave
  // where the P pieces of K-tile it+2 are issued: k-step 3 of tile it (after the hand-over), k-steps 0 and 1 of tile it+1.
  // 64 pieces issued by 8 waves at the same moment queue up behind one address unit (~50 cycles each, all waves blocked);
  // spread over the tile each one costs ~18 cycles and hides under an MFMA.
  constexpr int N3 = (P + 2) / 3;
  constexpr int N0 = (P + 1) / 3;
  constexpr int N1 = P - N3 - N0;
  bool pending = false;                  // pieces of the cursor's K-tile still to be issued
  const int xp = g.stagger;   // timing experiments only (results are wrong): 1 = no DMA wait, 2 = no DMA issue in the K loop, 4 = no fragment reads in the K loop
  auto issue_piece = [&](uint32_t base, auto p_c) {   // base = LDS byte offset of the target stage
    constexpr int p = decltype(p_c)::value;
    const uint32_t sa = a_soff + i_k * (BK * 2), sb = b_soff + i_k * (BK * 2), lb = base;
    if constexpr (p < A_INSTR) {
      if constexpr ((p & 3) == 0) vitx_dma16(rsA, lds_wa + lb + p * 1024, offA[p], sa);
      else vitx_dma16_cont<(p & 3) * 1024>(rsA, offA[p], sa);
    } else {
      constexpr int q = p - A_INSTR;
      if constexpr ((q & 3) == 0) vitx_dma16(rsB, lds_wb + lb + q * 1024, offB[q], sb);
      else vitx_dma16_cont<(q & 3) * 1024>(rsB, offB[q], sb);
    }
  };
  auto i_advance = [&]() {
    ++issued;
    if (++i_k == nk) {
      i_k = 0;
      i_logical += w_stride;
      i_more = persistent && i_logical < w_limit;
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
  // (Measured and removed in round 4: pieces one behind every SP-th MFMA with per-wave phases, fragment reads one per MFMA, the two-barriers-per-
  //  k-step anti-phase schedule, the wave-group ping-pong kernel -- all within +-3 % of this schedule on every shape, profiles/r3/gemm_sweep_*.)
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
  constexpr bool GTAB = MODE == EPI_BIAS_GELU && BM == 256;   // GELU by table: 15 KiB behind the two pipeline buffers (the 320-row tile has no room)
  if constexpr (GTAB) {
    if (gelu_tab != nullptr) {
      uint32_t* t = (uint32_t*)(smem + 2 * STAGE);
      for (int i = tid; i < 2 * GT_NE; i += NW * 64) t[i] = gelu_tab[i];   // visible to every wave behind the hand-over barrier below
    }
  }
  handover();
  load_frags(fa[0], fb[0], smem, 0);
  int it = 0;   // consumed K-tile counter of the stream
  int tile_idx = 0;
#ifndef VITX_GEMM_STAMPS_BUILD
#define VITX_GEMM_STAMPS_BUILD 0
#endif
  constexpr bool kStamps = VITX_GEMM_STAMPS_BUILD != 0;   // tools/build_variant.sh stamps "-DVITX_GEMM_STAMPS_BUILD=1" gemm_bf16_pipe.hip; cycle stamps of the tile phases (diagnostic build only: the stores leave VMEM state pending across the K loop)
  auto stamp = [&](int k) {
    if constexpr (kStamps)
      if (g.stamps && tid == 0 && bid < 512 && tile_idx < 15) {
        g.stamps[((int64_t)bid * 16 + tile_idx) * 4 + k] = __builtin_readcyclecounter();
        // where this workgroup runs (HW_ID: CU / SE; XCC_ID), record 15 of its row: which workgroups share a CU is read from here
        if (tile_idx == 0 && k == 0)
          g.stamps[((int64_t)bid * 16 + 15) * 4] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
      }
  };
  // (the ONLY back edge of this loop runs through handover(): on any other path the compiler's scoreboard would carry the epilogue's
  //  bias / residual loads into the K loop as "pending" and protect their registers with vmcnt waits there -- see handover())
  for (int logical = w_first;; logical += w_stride, ++tile_idx) {
    int tile_m, tile_n;
    tile_of(logical, tile_m, tile_n);
    const bool has_next = persistent && logical + w_stride < w_limit;
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
          if (!(xp & 4)) load_frags(fa[CUR ^ 1], fb[CUR ^ 1], base, ks + 1);
        } else {
          // K-tile hand-over in front of the last k-step's MFMAs (ONE instruction stream for every case -- branching the MFMA
          // sequence makes the allocator copy accumulators): K-tile it+1 has landed, buffer it&1 is fully read by every wave.
          handover();
          mfma_range(ic<CUR>{}, ic<0>{}, ic<2>{});
          __builtin_amdgcn_sched_barrier(0);
          // first fragments of the next K-tile; at the last K-tile of an output tile they are loaded AFTER the epilogue instead
          // (kept live across it they cost the 320-row variants their last free registers)
          if (kt + 1 < nk && !(xp & 4)) load_frags(fa[0], fb[0], smem + ((it + 1) & 1) * STAGE, 0);
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
    bool wp_done = false;   // this tile went out through one of the fast forms
    {
      const bool interior = epilogue_fast_ok(ep, MODE) && (tile_m + 1) * BM <= ep.M && (tile_n + 1) * BN <= ep.N;
      // ---- WP, the wave-private epilogue (interior tiles of the four fused epilogues of the Dense layers).
      // Each wave drains its OWN 128 x 64 accumulator tile through its OWN 8 KiB of the free pipeline buffer -- the two 4-KiB areas its own DMA
      // pieces of the next refill land in (A rows 32 w .. 32 w + 31, B rows likewise) -- 32 rows at a time: accumulator layout in (a lane holds 4
      // columns of one row), row-contiguous layout out (bf16 outputs: 8 lanes x 16 B = one full 128-B line per row, 8 rows per store instruction;
      // fp32 outputs: 16 lanes x 16 B = 256 B per row, 4 rows per instruction).  LDS executes a wave's instructions in order, so the write-read
      // round trip needs no wait and NO workgroup barrier: the eight waves drift apart, one wave's LDS round trip / GELU arithmetic runs under
      // another wave's global loads and stores instead of all eight marching through eight write-barrier-read-store rounds in lock-step, and a
      // wave that is done refills its slice and starts the next tile's first K-tile (already in LDS) without waiting for the others (the next
      // workgroup barrier is that K-tile's hand-over).  Nobody else reads a wave's slice after the hand-over barrier of the last K-tile, and
      // nobody else writes it before the refill the wave issues itself.
      //   bf16 staging (plain store, GELU): a 32 x 64 block is 4 KiB -- area 0 / area 1 alternate (GELU: gelu' in area 0, gelu in area 1).
      //   fp32 staging (residual, GELU VJP): columns 0..31 in area 0, 32..63 in area 1 with rows and chunk parity flipped (row ^ 1, chunk ^ 1) so that
      //   the 16 lanes of a row-contiguous read hit 64 different banks; 16-B chunks are XOR-swizzled with (row >> 1) & 7 in both layouts.
      static_assert((BM == 256 || BM == 320 || BM == 192) && A_INSTR >= 4 && B_INSTR == 4 && WTN == 64 && MT * 32 == WTM && NT == 2, "wave-private epilogue: 256 / 320 x 256 tiles, 2 x 4 waves");
      if constexpr (MODE == EPI_STORE || MODE == EPI_BIAS_GELU || MODE == EPI_BIAS_RESID || MODE == EPI_GELU_BWD) {
        const bool wp_ok = interior && (MODE == EPI_BIAS_RESID || ep.wide_ok) && !(MODE == EPI_STORE && has_bias);
        if (wp_ok) {
          wp_done = true;
          char* const sa = smem + ((it + 1) & 1) * STAGE + wave * (A_INSTR * 1024);
          char* const sb = smem + ((it + 1) & 1) * STAGE + A_BYTES + wave * (B_INSTR * 1024);
          // every per-lane address below derives from `le`, the lane index RECOMPUTED here behind an asm the optimiser cannot see through: as loop
          // invariants of the persistent tile loop the addresses were hoisted in front of it and kept live across the K loop (78-107 registers
          // spilled to scratch -- and a scratch reload is a VMEM load the compiler then waits for INSIDE the K loop, where vmcnt also counts the DMA)
          const int le = wp_lane();
          const int m = le & 31, swm = (m >> 1) & 7, kh = le >> 5;
          const int row_w = tile_m * BM + wm * WTM, col_w = tile_n * BN + wn * WTN;
          auto to_bf16x4 = [](float a, float b, float c, float d) { bf16x4 o; o[0] = (bf16_t)a; o[1] = (bf16_t)b; o[2] = (bf16_t)c; o[3] = (bf16_t)d; return o; };
          // fp32 staging of accumulator block i0 (32 rows x 64 columns)
          auto stage_f32 = [&](auto i_c) {
            constexpr int i0 = decltype(i_c)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int ch = 2 * q + kh;
              *(float4*)(sa + m * 128 + ((ch ^ swm) << 4)) = make_float4(acc[i0][0][4 * q], acc[i0][0][4 * q + 1], acc[i0][0][4 * q + 2], acc[i0][0][4 * q + 3]);
              *(float4*)(sb + (m ^ 1) * 128 + ((ch ^ swm ^ 1) << 4)) = make_float4(acc[i0][1][4 * q], acc[i0][1][4 * q + 1], acc[i0][1][4 * q + 2], acc[i0][1][4 * q + 3]);
            }
          };
          if constexpr (MODE == EPI_STORE) {
            bf16_t* const outp = (bf16_t*)ep.out + out_off + (int64_t)row_w * ep.ldo + col_w;
            const uint32_t ldo = (uint32_t)ep.ldo, ooff = (uint32_t)(le >> 3) * ldo + (uint32_t)(le & 7) * 8u;
            auto run = [&](auto nt_c) {
              constexpr int POL = decltype(nt_c)::value;
              static_for<MT>([&](auto i_c) {
                constexpr int i0 = decltype(i_c)::value;
                char* const ar = (i0 & 1) ? sb : sa;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                  for (int q = 0; q < 4; ++q)
                    *(bf16x4*)(ar + m * 128 + (((j * 4 + q) ^ swm) << 4) + kh * 8) =
                        to_bf16x4(acc[i0][j][4 * q], acc[i0][j][4 * q + 1], acc[i0][j][4 * q + 2], acc[i0][j][4 * q + 3]);
                bf16x8 v[4];   // all four row groups are read before the first store: one LDS latency per block, not four in series
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const int r = k * 8 + (le >> 3);
                  v[k] = *(const bf16x8*)(ar + r * 128 + (((le & 7) ^ ((r >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) wp_store16<POL>(outp, ooff + (uint32_t)(i0 * 32 + k * 8) * ldo, v[k]);
              });
            };
            if (ep.nt_out) run(ic<1>{});
            else run(ic<0>{});
          } else if constexpr (MODE == EPI_BIAS_GELU) {
            // (the bias values of a 32-column block are re-read for every 32 x 32 accumulator block -- L1 hits -- instead of living in 32 registers
            //  through the epilogue: with them the table form spilled, and a scratch reload is a VMEM load the compiler waits for inside the K loop)
            const float* const bias_w = has_bias ? ep.bias + col_w + 4 * kh : nullptr;
            bf16_t* const o1 = (bf16_t*)ep.out + (int64_t)row_w * ep.ldo + col_w;
            bf16_t* const o2 = (bf16_t*)ep.out2 + (int64_t)row_w * ep.ldo2 + col_w;
            const uint32_t ldo = (uint32_t)ep.ldo, ldo2 = (uint32_t)ep.ldo2;
            const uint32_t o1off = (uint32_t)(le >> 3) * ldo + (uint32_t)(le & 7) * 8u, o2off = (uint32_t)(le >> 3) * ldo2 + (uint32_t)(le & 7) * 8u;
            const bool use_tab = GTAB && gelu_tab != nullptr;
            const char* const gt = smem + 2 * STAGE;
            auto run = [&](auto nt_c, auto tab_c) {
              constexpr int POL = decltype(nt_c)::value;
              constexpr bool TAB = decltype(tab_c)::value;
              static_for<MT>([&](auto i_c) {
                constexpr int i0 = decltype(i_c)::value;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                  float4 bzj[4];
#pragma unroll
                  for (int q = 0; q < 4; ++q) bzj[q] = bias_w ? *(const float4*)(bias_w + j * 32 + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    // GELU and its derivative of the pre-activation as bf16 would store it
                    const bf16x4 h = to_bf16x4(acc[i0][j][4 * q] + bzj[q].x, acc[i0][j][4 * q + 1] + bzj[q].y, acc[i0][j][4 * q + 2] + bzj[q].z,
                                               acc[i0][j][4 * q + 3] + bzj[q].w);
                    const int off = m * 128 + (((j * 4 + q) ^ swm) << 4) + kh * 8;
                    if constexpr (TAB) {
                      // (Phi as fp16 | gelu' as bf16) of each 16-bit pattern from the LDS table; gelu = x Phi
                      const uint32_t p0 = __builtin_bit_cast(uint2, h).x, p1 = __builtin_bit_cast(uint2, h).y;
                      auto look = [&](uint32_t bits15, uint32_t sign) {
                        const int t = min(max((int)bits15 - GT_LO, 0), GT_NE - 1);             // clamp to the table's binades (both ends are exact: see above)
                        return *(const uint32_t*)(gt + ((uint32_t)t + sign * GT_NE) * 4u);
                      };
                      const uint32_t e0 = look(p0 & 0x7fffu, (p0 >> 15) & 1u), e1 = look((p0 >> 16) & 0x7fffu, p0 >> 31);
                      const uint32_t e2 = look(p1 & 0x7fffu, (p1 >> 15) & 1u), e3 = look((p1 >> 16) & 0x7fffu, p1 >> 31);
                      auto phi = [](uint32_t e) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(e >> 16)); };
                      const float x0 = __builtin_bit_cast(float, p0 << 16), x1 = __builtin_bit_cast(float, p0 & 0xffff0000u);
                      const float x2 = __builtin_bit_cast(float, p1 << 16), x3 = __builtin_bit_cast(float, p1 & 0xffff0000u);
                      uint2 gdw;   // gelu' of the four columns: the low halves of the entries, packed
                      gdw.x = __builtin_amdgcn_perm(e1, e0, 0x05040100u);
                      gdw.y = __builtin_amdgcn_perm(e3, e2, 0x05040100u);
                      *(uint2*)(sa + off) = gdw;
                      *(bf16x4*)(sb + off) = to_bf16x4(x0 * phi(e0), x1 * phi(e1), x2 * phi(e2), x3 * phi(e3));
                    } else {
                      float4 g, gd;
                      gelu_both4(make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]), g, gd);
                      *(bf16x4*)(sa + off) = to_bf16x4(gd.x, gd.y, gd.z, gd.w);
                      *(bf16x4*)(sb + off) = to_bf16x4(g.x, g.y, g.z, g.w);
                    }
                  }
                }
                bf16x8 vd[4], vg[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const int r = k * 8 + (le >> 3);
                  const int off = r * 128 + (((le & 7) ^ ((r >> 1) & 7)) << 4);
                  vd[k] = *(const bf16x8*)(sa + off);
                  vg[k] = *(const bf16x8*)(sb + off);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  wp_store16<POL>(o1, o1off + (uint32_t)(i0 * 32 + k * 8) * ldo, vd[k]);
                  wp_store16<0>(o2, o2off + (uint32_t)(i0 * 32 + k * 8) * ldo2, vg[k]);
                }
              });
            };
            if constexpr (GTAB) {
              if (use_tab) { if (ep.nt_out) run(ic<1>{}, std::true_type{}); else run(ic<0>{}, std::true_type{}); }
              else if (ep.nt_out) run(ic<1>{}, std::false_type{});
              else run(ic<0>{}, std::false_type{});
            } else {
              if (ep.nt_out) run(ic<1>{}, std::false_type{});
              else run(ic<0>{}, std::false_type{});
            }
          } else if constexpr (MODE == EPI_BIAS_RESID) {
            // out[f32] = resid + (acc + bias) [* scale]; LayerScale (cait.py:47-48) also keeps f = acc + bias in out2 (epilogue_fast4's arithmetic).
            // Addresses: wave-uniform 64-bit bases + 32-bit lane offsets (a 64-bit multiply per row and lane cost the registers the rows need)
            const int c16 = le & 15, ar1 = c16 >> 3;
            const uint32_t lr = (uint32_t)(le >> 4), lc = (uint32_t)c16 * 4u;
            const float4 bb = has_bias ? *(const float4*)(ep.bias + col_w + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 ss = has_scale ? *(const float4*)(ep.scale + col_w + lc) : make_float4(1.f, 1.f, 1.f, 1.f);
            const char* const rb = ar1 ? sb : sa;
            const float* const rp = ep.resid + (int64_t)row_w * ep.ldr + col_w;
            float* const op = (float*)ep.out + (int64_t)row_w * ep.ldo + col_w;
            bf16_t* const o2 = (has_scale && ep.out2) ? (bf16_t*)ep.out2 + (int64_t)row_w * ep.ldo2 + col_w : nullptr;
            const uint32_t ldr = (uint32_t)ep.ldr, ldo = (uint32_t)ep.ldo, ldo2 = (uint32_t)ep.ldo2;
            const uint32_t roff = lr * ldr + lc, ooff = lr * ldo + lc, o2off = lr * ldo2 + lc;
            static_for<MT>([&](auto i_c) {
              constexpr int i0 = decltype(i_c)::value;
              // the 32 rows in steps of 16 (320-row tiles: 8 -- 160 accumulator registers leave room for less); the residual rows of a step are
              // requested before the LDS round trip / under the stores of the step before (a missing bias / scale is the exact identity: + 0, x 1 --
              // one instruction stream for the four combinations)
              constexpr int RS = BM == 320 ? 8 : 16, NR = RS / 4, NS = 32 / RS;
              float4 x[NR], v[NR];
#pragma unroll
              for (int k = 0; k < NR; ++k) x[k] = *(const float4*)(rp + (roff + (uint32_t)(i0 * 32 + k * 4) * ldr));
              stage_f32(i_c);
#pragma unroll
              for (int hf = 0; hf < NS; ++hf) {
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                  const int r = hf * RS + k * 4 + (int)lr;
                  v[k] = *(const float4*)(rb + (r ^ ar1) * 128 + ((((c16 & 7) ^ ((r >> 1) & 7)) ^ ar1) << 4));
                }
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                  const uint32_t rr = (uint32_t)(i0 * 32 + hf * RS + k * 4);
                  const float4 f = make_float4(v[k].x + bb.x, v[k].y + bb.y, v[k].z + bb.z, v[k].w + bb.w);
                  if (o2) st4<bf16_t>(o2 + (o2off + rr * ldo2), f);
                  const float4 o = make_float4(x[k].x + f.x * ss.x, x[k].y + f.y * ss.y, x[k].z + f.z * ss.z, x[k].w + f.w * ss.w);
                  if (hf + 1 < NS) x[k] = *(const float4*)(rp + (roff + (rr + (uint32_t)RS) * ldr));   // the next step's residual row, in flight under the stores
                  *(float4*)(op + (ooff + rr * ldo)) = o;
                }
              }
            });
          } else {   // EPI_GELU_BWD: out[bf16] = acc * stored gelu'(h), column sums of what was stored (epilogue_wide8's arithmetic)
            const int c8 = le & 7, ar1 = c8 >> 2, ch0 = (c8 & 3) * 2;
            const uint32_t lr = (uint32_t)(le >> 3), lc = (uint32_t)c8 * 8u;
            const char* const rb = ar1 ? sb : sa;
            const bf16_t* const xp = (const bf16_t*)ep.aux + (int64_t)row_w * ep.ldaux + col_w;
            bf16_t* const op = (bf16_t*)ep.out + (int64_t)row_w * ep.ldo + col_w;
            const uint32_t ldx = (uint32_t)ep.ldaux, ldo = (uint32_t)ep.ldo;
            const uint32_t xoff = lr * ldx + lc, ooff = lr * ldo + lc;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 cs = z4, cs2 = z4;   // column sums of what this lane stores (columns col_w + lc .. + 7 over the wave's 128 rows)
            static_for<MT>([&](auto i_c) {
              constexpr int i0 = decltype(i_c)::value;
              bf16x8 x[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) x[k] = *(const bf16x8*)(xp + (xoff + (uint32_t)(i0 * 32 + k * 8) * ldx));
              stage_f32(i_c);
              float4 va[4], vb[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int r = k * 8 + (int)lr;
                const char* rowp = rb + (r ^ ar1) * 128;
                const int swr = ((r >> 1) & 7) ^ ar1;
                va[k] = *(const float4*)(rowp + ((ch0 ^ swr) << 4));
                vb[k] = *(const float4*)(rowp + (((ch0 + 1) ^ swr) << 4));
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float4 a = va[k], b = vb[k];
                const bf16x8 o = pack_bf16x8(make_float4(a.x * (float)x[k][0], a.y * (float)x[k][1], a.z * (float)x[k][2], a.w * (float)x[k][3]),
                                             make_float4(b.x * (float)x[k][4], b.y * (float)x[k][5], b.z * (float)x[k][6], b.w * (float)x[k][7]));
                wp_store16<0>(op, ooff + (uint32_t)(i0 * 32 + k * 8) * ldo, o);
                cs.x += (float)o[0]; cs.y += (float)o[1]; cs.z += (float)o[2]; cs.w += (float)o[3];       // as stored (rounded to bf16)
                cs2.x += (float)o[4]; cs2.y += (float)o[5]; cs2.z += (float)o[6]; cs2.w += (float)o[7];
              }
            });
            if (ep.colsum != nullptr) {
              // the 8 lanes with the same le & 7 hold the same columns for different rows: fixed-order butterfly over lane bits 3, 4, 5; the wave's
              // partial row goes to colsum row 2 tile_m + wm (the reduction pass behind the launch adds the rows in order)
              auto red = [&](float v) { v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; };
              cs.x = red(cs.x); cs.y = red(cs.y); cs.z = red(cs.z); cs.w = red(cs.w);
              cs2.x = red(cs2.x); cs2.y = red(cs2.y); cs2.z = red(cs2.z); cs2.w = red(cs2.w);
              if (le < 8) {
                float* crow = ep.colsum + (int64_t)(tile_m * 2 + wm) * ep.ldcs + col_w + lc;
                *(float4*)crow = cs;
                *(float4*)(crow + 4) = cs2;
              }
            }
          }
        }
      }
      {
        if (!wp_done) {
          // Everything the fast forms above do not take (ragged edge tiles, unaligned leading dimensions, a plain store with a bias, the fp32 /
          // patch-embedding / split-K epilogues): the same wave-private fp32 staging, then the GENERIC epilogue functor on row-contiguous float4
          // pieces (16 lanes x 16 B per row) with every bounds test per element.
          const int le = wp_lane();
          char* const sa = smem + ((it + 1) & 1) * STAGE + wave * (A_INSTR * 1024);
          char* const sb = smem + ((it + 1) & 1) * STAGE + A_BYTES + wave * (B_INSTR * 1024);
          const int m = le & 31, swm = (m >> 1) & 7, kh = le >> 5;
          const int row_w = tile_m * BM + wm * WTM, col_w = tile_n * BN + wn * WTN;
          const int c16 = le & 15, ar1 = c16 >> 3, lr = le >> 4, gcolr = col_w + c16 * 4;
          const char* const rb = ar1 ? sb : sa;
          float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
          static_for<MT>([&](auto i_c) {
            constexpr int i0 = decltype(i_c)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int ch = 2 * q + kh;
              *(float4*)(sa + m * 128 + ((ch ^ swm) << 4)) = make_float4(acc[i0][0][4 * q], acc[i0][0][4 * q + 1], acc[i0][0][4 * q + 2], acc[i0][0][4 * q + 3]);
              *(float4*)(sb + (m ^ 1) * 128 + ((ch ^ swm ^ 1) << 4)) = make_float4(acc[i0][1][4 * q], acc[i0][1][4 * q + 1], acc[i0][1][4 * q + 2], acc[i0][1][4 * q + 3]);
            }
#pragma unroll 2
            for (int k = 0; k < 8; ++k) {
              const int r = k * 4 + lr;
              const float4 v = *(const float4*)(rb + (r ^ ar1) * 128 + ((((c16 & 7) ^ ((r >> 1) & 7)) ^ ar1) << 4));
              const float4 res = epilogue_apply4<MODE, bf16_t>(ep, row_w + i0 * 32 + r, gcolr, v, out_off);
              if (MODE == EPI_GELU_BWD) { cs.x += res.x; cs.y += res.y; cs.z += res.z; cs.w += res.w; }
            }
          });
          if (MODE == EPI_GELU_BWD && ep.colsum != nullptr) {   // lanes 16 apart hold the same columns for different rows
            auto red = [&](float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; };
            cs.x = red(cs.x); cs.y = red(cs.y); cs.z = red(cs.z); cs.w = red(cs.w);
            if (le < 16 && gcolr < ep.N) {
              float* crow = ep.colsum + (int64_t)(tile_m * 2 + wm) * ep.ldcs + gcolr;
              if (gcolr + 3 < ep.N) *(float4*)crow = cs;
              else { crow[0] = cs.x; if (gcolr + 1 < ep.N) crow[1] = cs.y; if (gcolr + 2 < ep.N) crow[2] = cs.z; }
            }
          }
        }
      }

    }
    stamp(2);
    // (every path out of the epilogue has waited for its loads -- see S_WIDE / S_NARROW above; the structurizer routes the `break` through the
    //  block that is also the loop latch, so a path that left loads pending would show up as waits at the loop header)
    if (!has_next) break;
    // this wave's staging reads are done; NO workgroup barrier: its staging slice is the target of its OWN refill pieces only (no DMA piece is in
    // flight here: the refill is issued below, the prefetched K-tile landed before the epilogue)
    __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0)
    asm volatile("" ::: "memory");
    issue();                                      // deferred refill of the staging buffer: stream item it+1
    load_frags(fa[0], fb[0], smem + (it & 1) * STAGE, 0);
    stamp(3);
  }
}

template <int BM, int BN, int WM, int WN, int MODE>
void launch_pipe(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s, bool one_tile_per_wg = false) {
  constexpr int SMEM = 2 * (BM + BN) * BK * 2 + ((MODE == EPI_BIAS_GELU && BM == 256) ? GT_BYTES : 0);
  // buffer-addressed DMA: 31-bit byte offsets inside each operand; larger operands take the flat-addressed persistent kernel
  if (((int64_t)ceil_div(g.M, BM) * BM * g.lda + g.K) * 2 >= (1LL << 31) || ((int64_t)ceil_div(g.N, BN) * BN * g.ldb + g.K) * 2 >= (1LL << 31)) {
    launch_gemm_bf16_persistent_lockstep(BM, MODE, g, ep, s);
    return;
  }
  auto kern = gemm_bf16_nt_pipe_kernel<BM, BN, WM, WN, MODE>;
  vitx_set_max_smem((const void*)kern, SMEM);
  const int tiles_m = (int)ceil_div(g.M, BM), tiles_n = (int)ceil_div(g.N, BN);
  const int nk = g.K / BK;
  const int split = g.split_k > 1 ? g.split_k : 1;
  const int per = (int)ceil_div(nk, split);
  const int zs = (int)ceil_div(nk, per);
  static const int phase_env = [] { const char* v = vitx_env("VITX_GEMM_PHASE"); return v ? atoi(v) : 0; }();
  Bf16GemmArgs gp = g;
  if (gp.phase == 0) gp.phase = phase_env;
  static const int walk_env = [] { const char* v = vitx_env("VITX_GEMM_WALK"); return v ? atoi(v) : -1; }();   // A/B: 0 = interleaved chunks everywhere
  // (4-wave tiles: 80 KiB of LDS and <= 256 registers per wave, i.e. TWO workgroups per CU -- one's epilogue runs under the other's K loop)
  static const int grid_cap = [] { const char* v = vitx_env("VITX_GEMM_GRID"); return v ? atoi(v) : 256 * (WM * WN == 4 ? 2 : 1); }();   // experiment: fewer persistent workgroups
  // Beside collectives (data parallel: RCCL's workgroups hold CUs for as long as a collective runs) a persistent grid with static tile lists waits
  // for the workgroups that could not be placed; one tile per workgroup lets the hardware dispatcher balance.  Same kernel: a workgroup whose tile
  // list has one entry simply never takes the cross-tile path.
  const unsigned gx = (zs == 1 && !gemm_bf16_shared_gpu() && !g.shared_gpu && !one_tile_per_wg) ? (unsigned)std::min(tiles_m * tiles_n, grid_cap) : (unsigned)(tiles_m * tiles_n);
  dim3 grid(gx, (unsigned)zs), block(WM * WN * 64);
  // XCD-owned row bands (walk 2): wide outputs on the full persistent grid with enough row tiles for 8-row bands per XCD
  const bool walk2_ok = BM == 256 && zs == 1 && gx == (unsigned)grid_cap && (gx & 7) == 0 && gx < (unsigned)(tiles_m * tiles_n) && tiles_n >= 8 && tiles_m >= 64 && g.stagger != 8;
  gp.walk = (walk2_ok && walk_env != 0) ? 2 : 0;
  static const int gelu_table_on = [] { const char* v = vitx_env("VITX_GELU_TABLE"); return v ? atoi(v) : 1; }();   // 0: the polynomial form everywhere (A/B)
  const uint32_t* gtab = (MODE == EPI_BIAS_GELU && BM == 256 && gelu_table_on) ? gelu_table_dev() : nullptr;
  hipLaunchKernelGGL(kern, grid, block, SMEM, s, gp, ep, tiles_m, tiles_n, per, gtab);
}

template <int MODE>
void pipe_mode(int variant, const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  switch (variant) {
    case 9: launch_pipe<192, 128, 2, 2, MODE>(g, ep, s); break;          // two 4-wave workgroups per CU (80 KiB of LDS each), persistent: 512 workgroups
    case 10: launch_pipe<192, 128, 2, 2, MODE>(g, ep, s, true); break;   // the same, one tile per workgroup (the hardware dispatcher balances the tail)
    case 11: launch_pipe<320, 256, 2, 4, MODE>(g, ep, s); break;   // fewer, taller tiles: 768-wide outputs (474 instead of 591 tiles at 50k rows: 1.85 rounds of 256 CUs)
    default: launch_pipe<256, 256, 2, 4, MODE>(g, ep, s); break;   // 13
  }
}

}  // namespace

void launch_gemm_bf16_pipe(int variant, int mode, const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s) {
  switch (mode) {
    case EPI_STORE: pipe_mode<EPI_STORE>(variant, g, ep, s); break;
    case EPI_STORE_F32: pipe_mode<EPI_STORE_F32>(variant, g, ep, s); break;
    case EPI_BIAS_GELU: pipe_mode<EPI_BIAS_GELU>(variant, g, ep, s); break;
    case EPI_BIAS_RESID: pipe_mode<EPI_BIAS_RESID>(variant, g, ep, s); break;
    case EPI_PATCH: pipe_mode<EPI_PATCH>(variant, g, ep, s); break;
    case EPI_GELU_BWD: pipe_mode<EPI_GELU_BWD>(variant, g, ep, s); break;
    case EPI_PARTIAL: pipe_mode<EPI_PARTIAL>(variant, g, ep, s); break;
    default: break;
  }
}
