// Split-operand GEMM for the BF16X3 compute mode: fp32 operands in memory, fp32 accumulators, the products on the bf16 matrix pipe.
//   C[z][m][n] = alpha * sum_k A[z](m,k) * B[z](k,n)   (+ the shared fused epilogues; same interface as gemm_f32_mfma.hip / gemm_generic.hip)
// Every operand element x is split on its way into LDS into  hi = bf16(x)  and  lo = bf16(x - hi)  (x = hi + lo up to 2^-17 |x|), and every
// k-step issues THREE v_mfma_f32_32x32x16_bf16 per output block:  acc += a_hi b_lo;  acc += a_lo b_hi;  acc += a_hi b_hi  -- the a_lo b_lo term
// (<= 2^-16 of the product) is dropped.  Per product the relative error is ~2^-16 against bf16's 2^-8 and fp32's 2^-24: the Dense layers of
// vit.py:39,42,59,63,143,156 and their VJPs keep the 1e-3 parity gate of the fp32 mode at a third of the bf16 MFMA rate instead of the fp32
// pipe's sixteenth (157 TFLOP/s peak).  Everything around the GEMMs (LayerNorm, softmax, GELU, residual stream, gradients) is the fp32 mode's.
//
//
// gfx950 mapping: 128 x 128 x 32 tiles, 4 waves (2 x 2), 64 x 64 per wave = 2 x 2 MFMA blocks.  global fp32 (16-B loads) -> registers -> split ->
// LDS as four bf16 planes per buffer (A hi, A lo, B hi, B lo).  The plane layout follows the operand's unit-stride axis (template parameters):
//   K-contiguous source (activations of the forward / dgrad GEMMs, W^T of the dgrad):  [128 rows][32 k], 64-B rows whose 16-B chunks are swizzled
//     chunk ^= (row >> 2) & 3  (the 16 rows of a ds_read_b128 phase hit 16 different 4-bank groups); a fragment is ONE ds_read_b128;
//   row-contiguous source (the [in][out] weights of the forward, both operands of the weight gradient):  [32 k][128 rows], 256-B rows whose
//     32-B granules are swizzled  granule ^= 2 (k & 3); a fragment is TWO ds_read_b64_tr_b16 (the hardware transpose read, as in gemm_bf16_tn.hip).
// Either way a thread writes the four elements of one 16-B global load as one 8-B LDS word per plane.  Operands are swapped (D^T = B^T A^T) as in
// the other GEMM kernels so that a lane ends up with four consecutive output columns of one row.  LDS is double buffered (64 KiB, two workgroups
// per CU); the global loads of K-tile t+1 are in flight while tile t is multiplied.
#include "kernels.h"

namespace {

constexpr int XM = 128, XN = 128, XK = 32;
constexpr int PLANE = XM * XK * 2;   // bytes

__device__ const float4 x3_zero16 = {0.f, 0.f, 0.f, 0.f};   // what a staging load outside the operand reads

__device__ __forceinline__ bf16x8 x3_tr_frag(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// LAY 1: source K-contiguous, LAY 2: source row-contiguous (see above)
template <int LAY>
struct X3Operand {
  // the four 16-B loads of this thread for one K-tile: (row, k) of the first element and the axis the four elements run along
  static __device__ __forceinline__ void coords(int t, int p, int& row, int& kk) {
    if (LAY == 1) { row = p * 32 + (t >> 3); kk = (t & 7) * 4; }         // 4 consecutive k of one row
    else { kk = p * 8 + (t >> 5); row = (t & 31) * 4; }                  // 4 consecutive rows at one k
  }
  static __device__ __forceinline__ int lds_off(int row, int kk) {        // byte offset of the thread's 8-B word inside a plane
    if (LAY == 1) return row * 64 + ((((kk >> 3) ^ (row >> 2)) & 3) << 4) + (kk & 7) * 2;
    const int fb = row * 2;
    return kk * 256 + ((((fb >> 5) ^ (2 * (kk & 3))) & 7) << 5) + (fb & 31);
  }
};

// (measured and removed: one LDS buffer + one register set, three workgroups per CU -- 26.45 vs 25.67 ms per ViT-B/16 step at batch 64)
template <int MODE, int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GenericGemmArgs g, EpiParams ep) {
  __shared__ __attribute__((aligned(16))) char smem[2][4][PLANE];
  const int z = blockIdx.z, zb = z / g.nh, zh = z - zb * g.nh;
  const float* A = (const float*)g.A + (int64_t)zb * g.sAb + (int64_t)zh * g.sAh;
  const float* B = (const float*)g.B + (int64_t)zb * g.sBb + (int64_t)zh * g.sBh;
  const int64_t out_off = (int64_t)zb * ep.out_batch_stride + (int64_t)zh * ep.out_head_stride;
  const int Kz = (g.k_last > 0 && zb == g.nb - 1) ? g.k_last : g.K;   // split-K over the batch index: the last slice may be shorter
  const int nkt = (Kz + XK - 1) / XK;
  // Workgroup ids go round-robin to the 8 XCDs (each with its own 4 MiB L2).  XCD x takes a CONTIGUOUS chunk of the tile list, and the list walks
  // bands of 8 row tiles column-major, so the ~64 workgroups an XCD runs at any time form an 8 x 8 block of tiles: 16 operand panels in that L2
  // instead of 64 + 1 (in plain row-major order over all XCDs every fp32 panel was fetched ~8x: the kernel ran at the 14 B/clk/CU of the L2-miss path).
  const int tiles_m = (g.M + XM - 1) / XM, tiles_n = (g.N + XN - 1) / XN;
  int tm, tn;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    constexpr int GM = 8;
    const int group = GM * tiles_n, gid = L / group, first = gid * GM;
    const int gsz = min(tiles_m - first, GM), w = L - gid * group;
    tn = w / gsz;
    tm = first + (w - tn * gsz);
  }
  const int m0 = tm * XM, n0 = tn * XN;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // Staging addresses: one pointer and one byte step per 16-B load, advanced every K-tile.  A load whose rows lie beyond M / N is parked on a 16-B
  // zero word with step 0, and a load whose k lies beyond K (the last, partial K-tile; every load of the tiles behind the last one) is redirected
  // to the zero word by a select: the K loop has no predicated loads and no branches.  (Measured alternatives, both slower on MI355X: a uniform tile
  // pointer + 32-bit lane offsets with a separate edge path, 202 vs 237 TFLOP/s; buffer loads with hardware range checking.)
  // (pointers in the GLOBAL address space: as generic pointers the loads became flat_load, which also counts on lgkmcnt -- every wait for a
  //  fragment read then waited for the staging loads as well)
  typedef const __attribute__((address_space(1))) char* gptr;
  typedef const __attribute__((address_space(1))) f32x4* gptr4;
  gptr pa[4];
  gptr pb[4];
  int sta[4], stb[4], kka[4], kkb[4];
  const gptr zero16 = (gptr)(const char*)&x3_zero16;
  auto lane_ptrs = [&](auto lay_c, const float* P, int64_t s_idx, int64_t s_k, int idx0, int lim, gptr (&ptr)[4], int (&st)[4], int (&kks)[4]) {
    constexpr int LAY = decltype(lay_c)::value;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int row, kk;
      X3Operand<LAY>::coords(t, p, row, kk);
      const bool ok = idx0 + row < lim;
      const float* q = LAY == 1 ? P + (int64_t)(idx0 + row) * s_idx + kk : P + (int64_t)kk * s_k + (idx0 + row);
      ptr[p] = ok ? (gptr)(const char*)q : zero16;
      st[p] = ok ? (int)((LAY == 1 ? (int64_t)XK : (int64_t)XK * s_k) * 4) : 0;
      kks[p] = kk;
    }
  };
  lane_ptrs(std::integral_constant<int, LA>{}, A, g.sam, g.sak, m0, g.M, pa, sta, kka);
  lane_ptrs(std::integral_constant<int, LB>{}, B, g.sbn, g.sbk, n0, g.N, pb, stb, kkb);
  f32x4 ra[4], rb[4];
  auto load_tile = [&](auto check_c, int k0, gptr (&ptr)[4], const int (&st)[4], const int (&kks)[4], f32x4 (&r)[4]) {   // K-tile at k0; advances the pointers
    constexpr bool CHECK = decltype(check_c)::value;   // false: the caller knows that the whole K-tile lies inside K
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      r[p] = *(gptr4)((!CHECK || k0 + kks[p] < Kz) ? ptr[p] : zero16);
      ptr[p] += st[p];
    }
  };
  int sa_off, sb_off;   // LDS word offset of load 0; load p sits p * 2048 bytes further in both layouts
  {
    int row, kk;
    X3Operand<LA>::coords(t, 0, row, kk);
    sa_off = X3Operand<LA>::lds_off(row, kk);
    X3Operand<LB>::coords(t, 0, row, kk);
    sb_off = X3Operand<LB>::lds_off(row, kk);
  }
  auto store_tile = [&](char* hi, char* lo, int off0, const f32x4 (&r)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      // per PAIR of elements: one packed convert (hi), two bit operations (hi back to fp32), one packed subtract, one packed convert (lo)
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const f32x2 x = {r[p][e], r[p][e + 1]};
        const bf16x2 hp = __builtin_convertvector(x, bf16x2);
        const uint32_t hb = __builtin_bit_cast(uint32_t, hp);
        const f32x2 hf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
        const bf16x2 lp = __builtin_convertvector(x - hf, bf16x2);
        h[e] = hp[0]; h[e + 1] = hp[1];
        l[e] = lp[0]; l[e + 1] = lp[1];
      }
      *(bf16x4*)(hi + off0 + p * 2048) = h;
      *(bf16x4*)(lo + off0 + p * 2048) = l;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int col = lane & 31, kh = lane >> 5;
  // transpose-read lane roles (LAY 2): lane = 16 G + q; q >> 2 -> k row within a group of 4, (G & 1) * 32 + (q & 3) * 8 -> byte inside the 64-B span
  const int q16 = lane & 15, trow = q16 >> 2, piece = ((lane >> 4) & 1) * 32 + (q16 & 3) * 8;
  auto frag = [&](auto lay_c, const char* plane, int row0, int ks) -> bf16x8 {   // rows row0 .. row0 + 31, k-step ks (16 wide)
    constexpr int LAY = decltype(lay_c)::value;
    if constexpr (LAY == 1) {
      const int row = row0 + col;
      return *(const bf16x8*)(plane + row * 64 + ((((ks * 2 + kh) ^ (row >> 2)) & 3) << 4));
    } else {
      const int k0r = 16 * ks + 8 * kh + trow, k1r = k0r + 4, fb = row0 * 2 + piece;
      return x3_tr_frag(plane + k0r * 256 + ((((fb >> 5) ^ (2 * (k0r & 3))) & 7) << 5) + (fb & 31),
                        plane + k1r * 256 + ((((fb >> 5) ^ (2 * (k1r & 3))) & 7) << 5) + (fb & 31));
    }
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < XK / 16; ++ks) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = frag(std::integral_constant<int, LA>{}, smem[buf][0], wm * 64 + i * 32, ks);
        al[i] = frag(std::integral_constant<int, LA>{}, smem[buf][1], wm * 64 + i * 32, ks);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = frag(std::integral_constant<int, LB>{}, smem[buf][2], wn * 64 + j * 32, ks);
        bl[j] = frag(std::integral_constant<int, LB>{}, smem[buf][3], wn * 64 + j * 32, ks);
      }
      // small terms first, the hi x hi term last
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    }
  };
  // Software pipeline over K-tiles, two register sets and two LDS buffers:  iteration t multiplies tile t out of buffer t & 1 while it converts and
  // writes tile t+1 (loaded two iterations ago) into the other buffer -- free since the barrier at the top of the iteration -- and then issues the loads
  // of tile t+3 into the register set just emptied.  Split and multiply are in ONE basic block and the scheduler is asked to alternate them (one MFMA,
  // a few VALU instructions): a wave issues in order, so a block of conversions in front of a block of MFMAs leaves the matrix pipe idle for the
  // first and the vector ALU idle for the second (that form: 230 TFLOP/s, MFMA pipe ~33 % busy with two workgroups per CU taking turns).
  using chk = std::true_type;
  using nochk = std::false_type;
  f32x4 ra1[4], rb1[4];
  load_tile(chk{}, 0, pa, sta, kka, ra);
  load_tile(chk{}, 0, pb, stb, kkb, rb);
  load_tile(chk{}, XK, pa, sta, kka, ra1);
  load_tile(chk{}, XK, pb, stb, kkb, rb1);
  store_tile(smem[0][0], smem[0][1], sa_off, ra);
  store_tile(smem[0][2], smem[0][3], sb_off, rb);
  load_tile(chk{}, 2 * XK, pa, sta, kka, ra);
  load_tile(chk{}, 2 * XK, pb, stb, kkb, rb);
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // then up to five VALU instructions
    }
  };
  // one pair of K-tiles (kt from buffer 0 / set 1 -> buffer 1, kt + 1 from buffer 1 / set 0 -> buffer 0); CHECK = the prefetched tiles kt+3, kt+4 may
  // reach beyond K (the main loop runs without the per-load select, the last two pairs with it)
  auto pair = [&](auto check_c, int kt) {
    __syncthreads();           // tile kt visible in buffer 0; every wave finished reading buffer 1 (tile kt-1)
    compute(0);
    store_tile(smem[1][0], smem[1][1], sa_off, ra1);            // tile kt+1 (zeros behind the last K-tile)
    store_tile(smem[1][2], smem[1][3], sb_off, rb1);
    interleave();
    load_tile(check_c, (kt + 3) * XK, pa, sta, kka, ra1);
    load_tile(check_c, (kt + 3) * XK, pb, stb, kkb, rb1);
    if (kt + 1 < nkt) {
      __syncthreads();
      compute(1);
      store_tile(smem[0][0], smem[0][1], sa_off, ra);           // tile kt+2
      store_tile(smem[0][2], smem[0][3], sb_off, rb);
      interleave();
      load_tile(check_c, (kt + 4) * XK, pa, sta, kka, ra);
      load_tile(check_c, (kt + 4) * XK, pb, stb, kkb, rb);
    }
  };
  int kt = 0;
  for (; (kt + 5) * XK <= Kz; kt += 2) pair(nochk{}, kt);   // tiles kt+3 and kt+4 lie fully inside K
  for (; kt < nkt; kt += 2) pair(chk{}, kt);

  // lane: output row m = lane & 31 of each 32 x 32 block, columns 8q + 4 (lane >> 5) + {0..3}
  if (epilogue_fast_ok(ep, MODE) && m0 + XM <= ep.M && n0 + XN <= ep.N) {   // interior tile: the branch-free form of the bf16 kernels' epilogues
    auto run = [&](auto hb_c, auto hs_c) {
      constexpr bool HB = decltype(hb_c)::value, HS = decltype(hs_c)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = n0 + wn * 64 + j * 32 + 8 * q + 4 * kh;
          const float4 b4 = HB ? *(const float4*)(ep.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 s4 = HS ? *(const float4*)(ep.scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int row = m0 + wm * 64 + i * 32 + col;
            const float4 x = epilogue_fast_load<MODE, float>(ep, row, c);
            epilogue_fast4<MODE, float, HB, HS>(ep, row, c, make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]),
                                                b4, s4, x, out_off);
          }
        }
    };
    const bool has_bias = ep.bias != nullptr, has_scale = ep.scale != nullptr;
    if (has_bias && has_scale) run(std::true_type{}, std::true_type{});
    else if (has_bias) run(std::true_type{}, std::false_type{});
    else if (has_scale) run(std::false_type{}, std::true_type{});
    else run(std::false_type{}, std::false_type{});
    return;
  }
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

// 1: 16-B loads along k (k-contiguous rows that start 16-B aligned for every (batch, head)); 2: 16-B loads along m / n; 0: neither
int x3_layout(const GenericGemmArgs& g, const void* P, int64_t s_idx, int64_t s_k, int extent, int64_t sb, int64_t sh) {
  const bool strides_ok = sb % 4 == 0 && sh % 4 == 0 && al16(P);
  if (s_k == 1 && s_idx % 4 == 0 && g.K % 4 == 0 && strides_ok) return 1;
  if (s_idx == 1 && s_k % 4 == 0 && extent % 4 == 0 && strides_ok) return 2;
  return 0;
}

template <int LA, int LB>
void x3_launch(const GenericGemmArgs& g, const EpiParams& ep, int mode, hipStream_t s) {
  dim3 grid((unsigned)(ceil_div(g.N, XN) * ceil_div(g.M, XM)), 1, (unsigned)(g.nb * g.nh)), block(256);
#define VITX_CASE(MODE) case MODE: hipLaunchKernelGGL((gemm_bf16x3_kernel<MODE, LA, LB>), grid, block, 0, s, g, ep); break;
  switch (mode) {
    VITX_CASE(EPI_STORE) VITX_CASE(EPI_STORE_F32) VITX_CASE(EPI_BIAS_GELU) VITX_CASE(EPI_BIAS_RESID)
    VITX_CASE(EPI_PATCH) VITX_CASE(EPI_GELU_BWD) VITX_CASE(EPI_PARTIAL)
    default: break;
  }
#undef VITX_CASE
}

}  // namespace

// fp32 operands and outputs only, both operands readable with 16-B loads; anything else (small, oddly strided) stays on the exact kernels
bool gemm_bf16x3_supported(const GenericGemmArgs& g, int ta, int tb, int to) {
  if (!(ta == 0 && tb == 0 && to == 0 && g.M >= 64 && g.N >= 64 && g.K >= 16)) return false;
  return x3_layout(g, g.A, g.sam, g.sak, g.M, g.sAb, g.sAh) != 0 && x3_layout(g, g.B, g.sbn, g.sbk, g.N, g.sBb, g.sBh) != 0;
}

void launch_gemm_bf16x3(const GenericGemmArgs& g, const EpiParams& ep, int mode, hipStream_t s) {
  const int la = x3_layout(g, g.A, g.sam, g.sak, g.M, g.sAb, g.sAh), lb = x3_layout(g, g.B, g.sbn, g.sbk, g.N, g.sBb, g.sBh);
  if (la == 1 && lb == 1) x3_launch<1, 1>(g, ep, mode, s);        // dgrad: dY @ W^T
  else if (la == 1 && lb == 2) x3_launch<1, 2>(g, ep, mode, s);   // forward: X @ W
  else if (la == 2 && lb == 2) x3_launch<2, 2>(g, ep, mode, s);   // weight gradient: X^T @ dY
  else if (la == 2 && lb == 1) x3_launch<2, 1>(g, ep, mode, s);
}
