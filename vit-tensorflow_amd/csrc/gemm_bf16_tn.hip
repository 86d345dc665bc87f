// Weight-gradient ("TN") kernel of the bf16 MFMA GEMM family (see gemm_bf16.hip for the family's design notes).
#include "gemm_bf16_common.h"

namespace {

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
  // wave w moves token rows [w INSTR RPI, (w + 1) INSTR RPI) of each operand: its pieces are back to back in LDS, so that groups of four
  // share ONE M0 value and differ in the instruction's immediate offset (vitx_dma16_cont, common.h; the offset also shifts the global address)
  for (int i = 0; i < INSTR; ++i) {
    const int row = (wave * INSTR + i) * RPI + lane / LPR;
    const int pc = lane % LPR;
    const int c = ((((pc >> 1) ^ (2 * (row & 3))) << 1) | (pc & 1));   // logical 16-B chunk stored at physical chunk pc
    offA[i] = (uint32_t)(row * (int)g.lda + c * 8) * 2u - (uint32_t)((i & 3) * 1024);
    offB[i] = (uint32_t)(row * (int)g.ldb + c * 8) * 2u - (uint32_t)((i & 3) * 1024);
  }
  constexpr int P = 2 * INSTR;           // DMA pieces (1 KiB) per K-tile per wave: A pieces, then B pieces
  constexpr int Q = MT * NT;             // MFMAs per k-step per wave
  // pieces of K-tile kt+2 are issued in k-step 3 of tile kt (after the hand-over) and k-steps 0, 1 of tile kt+1 -- see the NT kernel
  constexpr int N3 = (P + 2) / 3, N0 = (P + 1) / 3, N1 = P - N3 - N0;
  // buffer-addressed DMA (see the NT pipe kernel): resource per operand, K-tile offset in the scalar offset
  // issued from inline asm (vitx_dma16, common.h): as builtins the compiler put `s_waitcnt vmcnt(0)` in front of EVERY k-step's transpose reads
  const i32x4 rsA = vitx_make_rsrc(Ag), rsB = vitx_make_rsrc(Bg);
  const uint32_t lds_w = vitx_lds_addr(smem) + (uint32_t)wave * (INSTR * 1024u);
  const uint32_t a_kstep = (uint32_t)(BK * g.lda * 2), b_kstep = (uint32_t)(BK * g.ldb * 2);   // bytes per K-tile (64 token rows)
  auto issue_piece = [&](int buf, int kt, auto p_c) {
    constexpr int p = decltype(p_c)::value;
    const uint32_t base = lds_w + (uint32_t)buf * STAGE;
    if constexpr (p < INSTR) {
      if constexpr ((p & 3) == 0) vitx_dma16(rsA, base + p * 1024, offA[p], (uint32_t)kt * a_kstep);
      else vitx_dma16_cont<(p & 3) * 1024>(rsA, offA[p], (uint32_t)kt * a_kstep);
    } else {
      constexpr int q = p - INSTR;
      if constexpr ((q & 3) == 0) vitx_dma16(rsB, base + OP_BYTES + q * 1024, offB[q], (uint32_t)kt * b_kstep);
      else vitx_dma16_cont<(q & 3) * 1024>(rsB, offB[q], (uint32_t)kt * b_kstep);
    }
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
  vitx_set_max_smem((const void*)kern, SMEM);
  const int tiles_m = (int)ceil_div(g.M, BT), tiles_n = (int)ceil_div(g.N, BT);
  const int nk = g.K / BK;
  const int split = g.split_k > 1 ? g.split_k : 1;
  const int per = (int)ceil_div(nk, split);
  const int zs = (int)ceil_div(nk, per);
  dim3 grid((unsigned)(tiles_m * tiles_n * zs)), block(WM * WN * 64);
  hipLaunchKernelGGL(kern, grid, block, SMEM, s, g, ep, tiles_m, tiles_n, per);
}

}  // namespace

// C[M=in][N=out] (split-K partials) = A[K=tokens][in]^T * B[K=tokens][out]; kernel: 1 = 128x128 tile, else 256x256
int gemm_bf16_tn_tile(int kernel, int M, int N) { kernel &= 15; return (kernel == 1 || (M <= 128 && N <= 128)) ? 128 : 256; }
void launch_gemm_bf16_tn(const Bf16GemmArgs& g0, const EpiParams& ep, hipStream_t s) {
  static const int xp = [] {
    const char* v = vitx_env("VITX_TN_XP");
    const int x = v ? atoi(v) : 0;
    if (x) fprintf(stderr, "[vitx] VITX_TN_XP=%d: timing experiment -- weight gradients are WRONG in this process\n", x);
    return x;
  }();
  Bf16GemmArgs g = g0;
  g.stagger = xp;
  {   // K slices are addressed through a buffer resource with 31-bit offsets (dense_wgrad picks the slice count accordingly): never launch beyond it
    const int nk = g.K / BK, split = g.split_k > 1 ? g.split_k : 1;
    const int64_t slice_rows = ceil_div(nk, split) * BK;
    if ((slice_rows * std::max(g.lda, g.ldb) + 256) * 2 >= (1LL << 31)) {
      // (no abort() from inside a library.)  dense_wgrad never gets here -- it raises the slice count or takes the transpose path; a caller
      // that does is told so and gets no launch (its output keeps whatever it held: the C ABI's gradient buffers are zero-initialised)
      fprintf(stderr, "[vitx] launch_gemm_bf16_tn: a K slice of %lld rows x %lld features exceeds the 2 GiB buffer range -- NOT launched; use more slices\n",
              (long long)slice_rows, (long long)std::max(g.lda, g.ldb));
      return;
    }
  }
  if (gemm_bf16_tn_tile(g.kernel, g.M, g.N) == 128) launch_tn_variant<128, 2, 2>(g, ep, s);
  else launch_tn_variant<256, 2, 4>(g, ep, s);
}
