// Fused multi-head self-attention for the bf16 path (vit.py:73-82: split -> QK^T*scale -> softmax -> AV
// -> merge heads) and its VJP.  The [b,h,n,n] score matrix is never written to HBM.
//
// Layout: packed qkv [b, n, 3, h, 64] bf16 exactly as the to_qkv Dense emits it (vit.py:72-74: the
// 'b n (h d) -> b h n d' rearranges become addressing), output o [b, n, h*64] (vit.py:82).
//
// gfx950 mapping (wave = 64, v_mfma_f32_16x16x32_bf16):
//   * one workgroup per (image, head); that head's K (row-major, XOR-swizzled 16-B chunks) and V
//     (transposed) live in LDS; each wave owns 16-query blocks.
//   * scores are computed TRANSPOSED (S^T = K Q^T) so a lane owns one query column: softmax row
//     reductions are in-lane + two xor-shuffles, and the bf16 P registers are directly the MFMA
//     B operand of O^T = V^T P^T.  The k-slot <-> key mapping of that second MFMA is a permutation
//     (slot (g,e) <-> key 32u + 4g + e, 16 + ...), applied identically to the V^T operand reads --
//     a contraction does not care about the order of its terms.
//   * backward = dQ kernel (wave per query block, same S^T layout) + dK/dV kernel (wave per 16-key
//     tile, S layout so a lane owns one key column); P is recomputed from the saved row LSE.
#include "kernels.h"
#include "attn_lds.h"

// tools/probe_attn.hip compiles this file with -DVITX_ATTN_PROBE: wave 0 of every workgroup then records 100-MHz timestamps at its
// phase boundaries (staging done / first pass / second pass / stores) -- the per-workgroup timeline behind DESIGN.md's attention notes.
#ifdef VITX_ATTN_PROBE
__device__ unsigned long long* vitx_attn_probe_buf;
#define ATTN_STAMP(i) do { if (threadIdx.x == 0 && vitx_attn_probe_buf) { vitx_attn_probe_buf[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); \
    if ((i) == 0) vitx_attn_probe_buf[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); } } while (0)   /* HW_ID | XCC_ID << 32 */
#else
#define ATTN_STAMP(i) do { } while (0)
#endif

namespace {

using namespace attn_lds;

constexpr int ATT_THREADS = 512;   // 8 waves share one head's LDS image (2 per SIMD)

// Where one (image, head) task finds its rows.  Layout 0 (the engine's): packed qkv [b, n, 3, h, 64] as the to_qkv Dense emits it and o / dO
// [b, n, h, 64].  Layout 1 ("planar", tools/probe_attn only so far): 3 h planes [b n][64] for q | k | v (and h planes for o / dO), i.e. every
// task's q, k, v, o are contiguous n x 128-B blocks -- the A/B that asks what the 128-B-segments-at-4.6-KB-stride access pattern costs.
struct AttnLayout {
  int64_t tok_stride;   // elements between consecutive tokens of one head's q (k, v) rows
  int64_t kv_off;       // q -> k and k -> v offset
  int64_t o_stride;     // elements between consecutive tokens of one head's o / dO rows
  int64_t q0, o0;       // offsets of this task's first q row / first o row
};
__device__ __forceinline__ AttnLayout attn_layout(int planar, int bi, int hi, int nb, int n, int h) {
  AttnLayout a;
  const int64_t inner = (int64_t)h * DH;
  if (planar) {
    const int64_t M = (int64_t)nb * n;
    a.tok_stride = DH; a.kv_off = (int64_t)h * M * DH; a.o_stride = DH;
    a.q0 = ((int64_t)hi * M + (int64_t)bi * n) * DH; a.o0 = a.q0;
  } else {
    a.tok_stride = 3 * inner; a.kv_off = inner; a.o_stride = inner;
    a.q0 = (int64_t)bi * n * a.tok_stride + hi * DH; a.o0 = (int64_t)bi * n * inner + hi * DH;
  }
  return a;
}

// rows [0, npad) of a [*, 64] bf16 matrix (row stride `stride` elements) -> LDS row-major swizzled image
// and/or transposed image T[dh][vs]; rows >= nvalid are zero-filled.
template <bool ROWMAJOR, bool TRANSPOSED>
__device__ __forceinline__ void stage_head(const bf16_t* src, int64_t stride, int nvalid, int npad, char* rm, bf16_t* tr, int vs,
                                           int tid, int nthreads) {
  for (int idx = tid; idx < npad * 8; idx += nthreads) {
    const int row = idx >> 3, c = idx & 7;
    bf16x8 v = zero8();
    if (row < nvalid) v = *(const bf16x8*)(src + (int64_t)row * stride + c * 8);
    if (ROWMAJOR) *(bf16x8*)(rm + row * ROWB + swz_chunk(row, c)) = v;
    if (TRANSPOSED) {
#pragma unroll
      for (int e = 0; e < 8; ++e) tr[(c * 8 + e) * vs + row] = v[e];
    }
  }
}

// (legacy) transposed-image operand: row dh, k-slots (g,e): e<4 -> col 32u+4g+e, e>=4 -> col 32u+16+4g+(e-4)
__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* tr, int vs, int dh, int u, int g) {
  const bf16x4 lo = *(const bf16x4*)(tr + dh * vs + 32 * u + 4 * g);
  const bf16x4 hi = *(const bf16x4*)(tr + dh * vs + 32 * u + 16 + 4 * g);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[4 + e] = hi[e]; }
  return r;
}
// ------------------------------------------------------------------------------------------ forward
// NTP = number of 16-key tiles (even); keys padded to 16*NTP.  A wave owns QB = 2 blocks of 16 queries at a time, so every
// K / V^T fragment read from LDS feeds two MFMAs (LDS bandwidth, not the matrix pipe, bounds these kernels).
constexpr int QB_FWD = 2;   // forward: 114 VGPRs, still two workgroups per CU
// (backward: one 16-row block per wave at a time -- two cost a workgroup of occupancy (146 / 212 VGPRs) and measured slower)

// One pair of 16-key tiles of the forward's second pass for QB query blocks.  These kernels are VALU-bound (rocprofv3 SQ counters:
// VALU ~53 % busy, the matrix pipe 23 %), so the body is written for instruction count: the four S^T chains (2 tiles x QB blocks) are
// issued chain-interleaved (no MFMA -> VALU wait states), the scale / shift and the row sums run as packed fp32 operations on register
// pairs, and the key mask exists only in the MASKED instantiation that the one pair reaching past n runs.
template <bool MASKED, int QB>
__device__ __forceinline__ void fwd_pair(const char* k_rm, const char* v_rm, int u, int qi, int g, int lane, int n, const bf16x8 (&qf)[QB][2],
                                         float sl2, const float (&m)[QB], f32x2 (&lv)[QB], f32x4 (&oacc)[QB][4]) {
  bf16x8 kf[2][2], vf[4];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) kf[tt][kh] = frag_rm(k_rm, (2 * u + tt) * 16 + qi, g + 4 * kh);
#pragma unroll
  for (int c = 0; c < 4; ++c) vf[c] = frag_trr(v_rm, c, u, lane);
  f32x4 a[2][QB];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int s = 0; s < QB; ++s) a[tt][s] = mfma16(kf[tt][0], qf[s][0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int s = 0; s < QB; ++s) a[tt][s] = mfma16(kf[tt][1], qf[s][1], a[tt][s]);
  bf16x8 pf[QB];
#pragma unroll
  for (int s = 0; s < QB; ++s) {
    f32x4 p[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const f32x4 e = a[tt][s] * sl2 - m[s];          // two packed fmas
#pragma unroll
      for (int r = 0; r < 4; ++r) p[tt][r] = fast_exp2(e[r]);
      if (MASKED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[tt][r] = ((2 * u + tt) * 16 + 4 * g + r) < n ? p[tt][r] : 0.f;
      }
      lv[s] += f32x2{p[tt][0], p[tt][1]} + f32x2{p[tt][2], p[tt][3]};
    }
    pf[s] = pack8(p[0], p[1]);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int s = 0; s < QB; ++s) oacc[s][c] = mfma16(vf[c], pf[s], oacc[s][c]);
}

// first pass: row maxima over one pair of tiles (MASKED: keys >= n count as -inf)
template <bool MASKED, int QB>
__device__ __forceinline__ void fwd_max_pair(const char* k_rm, int u, int qi, int g, int n, const bf16x8 (&qf)[QB][2], float (&m)[QB]) {
  bf16x8 kf[2][2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) kf[tt][kh] = frag_rm(k_rm, (2 * u + tt) * 16 + qi, g + 4 * kh);
  f32x4 a[2][QB];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int s = 0; s < QB; ++s) a[tt][s] = mfma16(kf[tt][0], qf[s][0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int s = 0; s < QB; ++s) a[tt][s] = mfma16(kf[tt][1], qf[s][1], a[tt][s]);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      f32x4 v = a[tt][s];
      if (MASKED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((2 * u + tt) * 16 + 4 * g + r) < n ? v[r] : -INFINITY;
      }
      m[s] = fmaxf(fmaxf(m[s], fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
}

template <int NTP>
__global__ __launch_bounds__(ATT_THREADS, 4) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                       int n, int h, float scale, const bf16_t* __restrict__ zero_page, int reverse, int planar) {
  constexpr int NKP = 16 * NTP, QB = QB_FWD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_rm = smem;                                   // [NKP][128 B] swizzled
  char* v_rm = smem + NKP * ROWB;                      // [NKP][128 B] swizzled (read through the hardware transpose)
  const int bh = reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x, bi = bh / h, hi = bh - bi * h;   // reverse: newest qkv rows first (memory-side cache)
  const AttnLayout L = attn_layout(planar, bi, hi, (int)gridDim.x / h, n, h);
  const int64_t tok_stride = L.tok_stride;
  const bf16_t* qbase = qkv + L.q0;
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ATTN_STAMP(0);
  stage_head_dma(qbase + L.kv_off, tok_stride, n, NKP, k_rm, zero_page, wave, lane, nwaves);
  stage_head_dma(qbase + 2 * L.kv_off, tok_stride, n, NKP, v_rm, zero_page, wave, lane, nwaves);

  const int qi = lane & 15, g = lane >> 4;
  const int nqb = (n + 16 * QB - 1) / (16 * QB);
  const float sl2 = scale * 1.44269504088896340736f;
  bf16x8 qf[QB][2];
  auto load_q = [&](int qb) {                        // rows >= n of the last block compute on the clamped row n-1 and are never stored
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      const int qc = min((qb * QB + s) * 16 + qi, n - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[s][ks] = *(const bf16x8*)(qbase + (int64_t)qc * tok_stride + (g + 4 * ks) * 8);
    }
  };
  if (wave < nqb) load_q(wave);                      // in flight together with the K / V images
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATTN_STAMP(1);
  __syncthreads();
  ATTN_STAMP(2);

  const int u_full = n >> 5;                        // tile pairs [0, u_full) hold valid keys only
  const int u_end = min(NTP / 2, (n + 31) >> 5);    // at most one more pair holds any valid key
  for (int qb = wave; qb < nqb; qb += nwaves) {
    // pass 1: row maxima of the scores.  S^T tile t: lane holds S[query qi][key 16t + 4g + r].
    float m[QB];
#pragma unroll
    for (int s = 0; s < QB; ++s) m[s] = -INFINITY;
#pragma unroll 1
    for (int u = 0; u < u_full; ++u) fwd_max_pair<false, QB>(k_rm, u, qi, g, n, qf, m);
    if (u_full < u_end) fwd_max_pair<true, QB>(k_rm, u_full, qi, g, n, qf, m);
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      m[s] = fmaxf(m[s], __shfl_xor(m[s], 16, 64));
      m[s] = fmaxf(m[s], __shfl_xor(m[s], 32, 64));
      m[s] *= sl2;
    }
    ATTN_STAMP(3);
    // pass 2: recompute the tile pair, p = 2^(s - m), accumulate the row sum and O^T += V^T P^T (unnormalised)
    f32x2 lv[QB];
    f32x4 oacc[QB][4];
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      lv[s] = f32x2{0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) oacc[s][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int u = 0; u < u_full; ++u) fwd_pair<false, QB>(k_rm, v_rm, u, qi, g, lane, n, qf, sl2, m, lv, oacc);
    if (u_full < u_end) fwd_pair<true, QB>(k_rm, v_rm, u_full, qi, g, lane, n, qf, sl2, m, lv, oacc);
    ATTN_STAMP(4);
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      const int q = (qb * QB + s) * 16 + qi;
      float ls = lv[s][0] + lv[s][1];
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      const float inv_l = 1.0f / ls;
      if (g == 0 && q < n) lse[(int64_t)bh * n + q] = (m[s] + log2f(ls)) * 0.69314718055994530942f;  // natural-log LSE
      if (q < n) {
        bf16_t* op = o + L.o0 + (int64_t)q * L.o_stride;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          bf16x4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(oacc[s][c][r] * inv_l);
          *(bf16x4*)(op + 16 * c + 4 * g) = ov;
        }
      }
    }
    if (qb + nwaves < nqb) load_q(qb + nwaves);
  }
  ATTN_STAMP(5);
}

// ------------------------------------------------------------------------------------------ backward
// One 16-query block of the dQ pass (K, V images in LDS): D[q] = sum_d dO*O, then per pair of key tiles P = 2^(S*sl2 - lse),
// dS = P * (dP - D) * scale, dQ^T += K^T dS^T.  `lse2` = this lane's query's LSE in the log2 domain; returns D for the caller to keep.
// The key mask is a template flag: only the one tile pair reaching past n pays for the compares and selects (these loops are VALU-bound).
template <bool MASKED>
__device__ __forceinline__ void dq_pair(const char* k_rm, const char* v_rm, int u, int qi, int g, int lane, int n, const bf16x8 (&qf)[2],
                                        const bf16x8 (&dof)[2], float sl2, float lse2, float scale, float nds, f32x4 (&dq)[4]) {
  bf16x8 kf[2][2], vf[2][2], kt[4];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      kf[tt][kh] = frag_rm(k_rm, (2 * u + tt) * 16 + qi, g + 4 * kh);
      vf[tt][kh] = frag_rm(v_rm, (2 * u + tt) * 16 + qi, g + 4 * kh);
    }
#pragma unroll
  for (int c = 0; c < 4; ++c) kt[c] = frag_trr(k_rm, c, u, lane);
  f32x4 sa[2], dp[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    sa[tt] = mfma16(kf[tt][0], qf[0], f32x4{0.f, 0.f, 0.f, 0.f});
    dp[tt] = mfma16(vf[tt][0], dof[0], f32x4{0.f, 0.f, 0.f, 0.f});
  }
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    sa[tt] = mfma16(kf[tt][1], qf[1], sa[tt]);
    dp[tt] = mfma16(vf[tt][1], dof[1], dp[tt]);
  }
  f32x4 ds[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const f32x4 e = sa[tt] * sl2 - lse2;
    const f32x4 w = dp[tt] * scale + nds;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = fast_exp2(e[r]);
      if (MASKED) p = ((2 * u + tt) * 16 + 4 * g + r) < n ? p : 0.f;
      ds[tt][r] = p * w[r];
    }
  }
  const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
  for (int c = 0; c < 4; ++c) dq[c] = mfma16(kt[c], dsf, dq[c]);
}

// per-lane global operands of one 16-query block (its q, dO and O rows): loaded one block AHEAD of their use, so that the round trip
// (1.5-2 us under load, tools/probe_attn) runs under the previous block's tile loop instead of in front of this one's
struct DqOperands { bf16x8 qf[2], dof[2], of[2]; };
__device__ __forceinline__ void dq_load(DqOperands& x, const bf16_t* qbase, int64_t tok_stride, const bf16_t* o_rows, const bf16_t* do_rows, int64_t inner,
                                        int qb, int lane, int n) {
  const int qc = min(qb * 16 + (lane & 15), n - 1), g = lane >> 4;   // rows >= n compute on the clamped row n-1 (finite) and are never stored
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    x.qf[ks] = *(const bf16x8*)(qbase + (int64_t)qc * tok_stride + (g + 4 * ks) * 8);
    x.dof[ks] = *(const bf16x8*)(do_rows + (int64_t)qc * inner + (g + 4 * ks) * 8);
    x.of[ks] = *(const bf16x8*)(o_rows + (int64_t)qc * inner + (g + 4 * ks) * 8);
  }
}

template <int NTP>
__device__ __forceinline__ float dq_block(const char* k_rm, const char* v_rm, const DqOperands& x, int64_t tok_stride, bf16_t* dq_rows, int qb, int lane,
                                          int n, float scale, float lse2) {
  const int qi = lane & 15, g = lane >> 4;
  const int q = qb * 16 + qi;
  const float sl2 = scale * 1.44269504088896340736f;
  const bf16x8 (&qf)[2] = x.qf;
  const bf16x8 (&dof)[2] = x.dof;
  float dsum = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) dsum += (float)x.dof[ks][e] * (float)x.of[ks][e];
  dsum += __shfl_xor(dsum, 16, 64);
  dsum += __shfl_xor(dsum, 32, 64);   // D[q] = sum_d dO*O
  const float nds = -dsum * scale;
  f32x4 dq[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) dq[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int u_full = n >> 5, u_end = min(NTP / 2, (n + 31) >> 5);
#pragma unroll 1
  for (int u = 0; u < u_full; ++u) dq_pair<false>(k_rm, v_rm, u, qi, g, lane, n, qf, dof, sl2, lse2, scale, nds, dq);
  if (u_full < u_end) dq_pair<true>(k_rm, v_rm, u_full, qi, g, lane, n, qf, dof, sl2, lse2, scale, nds, dq);
  if (q < n) {
    bf16_t* out = dq_rows + (int64_t)q * tok_stride;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)dq[c][r];
      *(bf16x4*)(out + 16 * c + 4 * g) = ov;
    }
  }
  return dsum;
}

// One 16-key block of the dK / dV pass (Q, dO images in LDS; lse_s (log2 domain) and d_s per query in LDS, rows >= n: 0).
// No masks: query rows >= n are zero rows of q / dO with lse = D = 0, so their P = 1 meets dO = 0 and their dS = 1 * (0 - 0); key lanes >= n
// compute on the clamped key n-1 and are never stored.  Tile pairs past n are skipped.
struct DkvOperands { bf16x8 kf[2], vf[2]; };   // one 16-key block's k and v rows (the B operands of S and dP), loaded one block ahead as well
__device__ __forceinline__ void dkv_load(DkvOperands& x, const bf16_t* kbase, int64_t tok_stride, int64_t inner, int kb, int lane, int n) {
  const int kc = min(kb * 16 + (lane & 15), n - 1), g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    x.kf[ks] = *(const bf16x8*)(kbase + (int64_t)kc * tok_stride + (g + 4 * ks) * 8);
    x.vf[ks] = *(const bf16x8*)(kbase + inner + (int64_t)kc * tok_stride + (g + 4 * ks) * 8);
  }
}

template <int NTP>
__device__ __forceinline__ void dkv_block(const char* q_rm, const char* do_rm, const float* lse_s, const float* d_s, const DkvOperands& x,
                                          int64_t tok_stride, int64_t inner, bf16_t* dk_rows, int kb, int lane, int n, float scale) {
  const int ki = lane & 15, g = lane >> 4;
  const int key = kb * 16 + ki;
  const float sl2 = scale * 1.44269504088896340736f;
  const bf16x8 (&kf)[2] = x.kf;
  const bf16x8 (&vf)[2] = x.vf;
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { dk[c] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int u_end = min(NTP / 2, (n + 31) >> 5);
#pragma unroll 1
  for (int u = 0; u < u_end; ++u) {
    f32x4 pp[2], ds[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = 2 * u + tt;
      const bf16x8 qa0 = frag_rm(q_rm, t * 16 + ki, g), qa1 = frag_rm(q_rm, t * 16 + ki, g + 4);
      const bf16x8 da0 = frag_rm(do_rm, t * 16 + ki, g), da1 = frag_rm(do_rm, t * 16 + ki, g + 4);
      float lq[4], dd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { lq[r] = lse_s[t * 16 + 4 * g + r]; dd[r] = d_s[t * 16 + 4 * g + r]; }
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      sa = mfma16(qa0, kf[0], sa);     // S[query 16t+4g+r][key ki]
      sa = mfma16(qa1, kf[1], sa);
      dp = mfma16(da0, vf[0], dp);     // dP same layout
      dp = mfma16(da1, vf[1], dp);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp2(fmaf(sa[r], sl2, -lq[r]));
        pp[tt][r] = p;
        ds[tt][r] = p * ((dp[r] - dd[r]) * scale);
      }
    }
    const bf16x8 pf = pack8(pp[0], pp[1]), dsf = pack8(ds[0], ds[1]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8 a1 = frag_trr(do_rm, c, u, lane), a2 = frag_trr(q_rm, c, u, lane);
      dv[c] = mfma16(a1, pf, dv[c]);
      dk[c] = mfma16(a2, dsf, dk[c]);
    }
  }
  if (key < n) {
    bf16_t* dkp = dk_rows + (int64_t)key * tok_stride;
    bf16_t* dvp = dkp + inner;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x4 a, b2;
#pragma unroll
      for (int r = 0; r < 4; ++r) { a[r] = (bf16_t)dk[c][r]; b2[r] = (bf16_t)dv[c][r]; }
      *(bf16x4*)(dkp + 16 * c + 4 * g) = a;
      *(bf16x4*)(dvp + 16 * c + 4 * g) = b2;
    }
  }
}

// backward as two launches (VITX_ATTN_BWD_SPLIT=1: the A/B and bit-identity form): dQ (+ row sums D to HBM) ...
template <int NTP>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                          const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                          float* __restrict__ dsum, bf16_t* __restrict__ dqkv, int n, int h, float scale,
                                                          const bf16_t* __restrict__ zero_page) {
  constexpr int NKP = 16 * NTP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_rm = smem;
  char* v_rm = smem + NKP * ROWB;
  const int bh = blockIdx.x, bi = bh / h, hi = bh - bi * h;
  const int inner = h * DH;
  const int64_t tok_stride = 3 * (int64_t)inner;
  const bf16_t* qbase = qkv + (int64_t)bi * n * tok_stride + hi * DH;
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stage_head_dma(qbase + inner, tok_stride, n, NKP, k_rm, zero_page, wave, lane, nwaves);
  stage_head_dma(qbase + 2 * inner, tok_stride, n, NKP, v_rm, zero_page, wave, lane, nwaves);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int64_t orow0 = (int64_t)bi * n * inner + hi * DH;
  for (int qb = wave; qb < (n + 15) / 16; qb += nwaves) {
    const int qi = lane & 15, q = qb * 16 + qi, qc = min(q, n - 1);
    DqOperands x;
    dq_load(x, qbase, tok_stride, o + orow0, d_o + orow0, inner, qb, lane, n);
    const float d = dq_block<NTP>(k_rm, v_rm, x, tok_stride, dqkv + (int64_t)bi * n * tok_stride + hi * DH, qb, lane, n, scale,
                                  lse[(int64_t)bh * n + qc] * 1.44269504088896340736f);
    if ((lane >> 4) == 0 && q < n) dsum[(int64_t)bh * n + q] = d;
  }
}

// ... and dK, dV
template <int NTP>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           bf16_t* __restrict__ dqkv, int n, int h, float scale, const bf16_t* __restrict__ zero_page) {
  constexpr int NQP = 16 * NTP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* q_rm = smem;
  char* do_rm = smem + NQP * ROWB;
  float* lse_s = (float*)(smem + 2 * NQP * ROWB);   // [NQP] (log2 domain)
  float* d_s = lse_s + NQP;                   // [NQP]
  const int bh = blockIdx.x, bi = bh / h, hi = bh - bi * h;
  const int inner = h * DH;
  const int64_t tok_stride = 3 * (int64_t)inner;
  const bf16_t* qbase = qkv + (int64_t)bi * n * tok_stride + hi * DH;
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stage_head_dma(qbase, tok_stride, n, NQP, q_rm, zero_page, wave, lane, nwaves);
  stage_head_dma(d_o + (int64_t)bi * n * inner + hi * DH, inner, n, NQP, do_rm, zero_page, wave, lane, nwaves);
  for (int i = tid; i < NQP; i += blockDim.x) {
    lse_s[i] = i < n ? lse[(int64_t)bh * n + i] * 1.44269504088896340736f : 0.f;
    d_s[i] = i < n ? dsum[(int64_t)bh * n + i] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kb = wave; kb < (n + 15) / 16; kb += nwaves) {
    DkvOperands x;
    dkv_load(x, qbase + inner, tok_stride, inner, kb, lane, n);
    dkv_block<NTP>(q_rm, do_rm, lse_s, d_s, x, tok_stride, inner, dqkv + (int64_t)bi * n * tok_stride + inner + hi * DH, kb, lane, n, scale);
  }
}

// ------------------------------------------------------------------------------------------ backward, both phases in one launch
// Phase 1 is the dQ pass, phase 2 the dK / dV pass, run back to back by the SAME workgroup on the same two LDS buffers (K, V images,
// then Q, dO images).  The arithmetic is that of the two-launch form (same bits); what changes is the traffic: as two launches
// each pass fetched its head's q, k, v, dO from HBM (310 MB per pass at ViT-B/16, more than the 256-MB MALL holds between them),
// here phase 2 finds them in the L2 its own phase 1 pulled them through 10-20 us earlier, and the row sums D never leave LDS.
template <int NTP>
__global__ __launch_bounds__(ATT_THREADS, 4) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                             const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dqkv, int n, int h, float scale,
                                                             const bf16_t* __restrict__ zero_page, int planar) {
  constexpr int NKP = 16 * NTP, NQP = 16 * NTP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_rm = smem;                                 // phase 1: K, V      phase 2: Q, dO
  char* v_rm = smem + NKP * ROWB;
  char* q_rm = k_rm;
  char* do_rm = v_rm;
  float* lse_s = (float*)(smem + 2 * NKP * ROWB);    // [NQP] (log2 domain), rows >= n: 0
  float* d_s = lse_s + NQP;                          // [NQP] D[q] = sum_d dO O, rows >= n: 0
  const int bh = blockIdx.x, bi = bh / h, hi = bh - bi * h;
  const AttnLayout L = attn_layout(planar, bi, hi, (int)gridDim.x / h, n, h);
  const int64_t tok_stride = L.tok_stride, inner = L.kv_off, ostr = L.o_stride;
  const bf16_t* qbase = qkv + L.q0;
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stage_head_dma(qbase + inner, tok_stride, n, NKP, k_rm, zero_page, wave, lane, nwaves);
  stage_head_dma(qbase + 2 * inner, tok_stride, n, NKP, v_rm, zero_page, wave, lane, nwaves);
  for (int i = tid; i < NQP; i += blockDim.x) {
    lse_s[i] = i < n ? lse[(int64_t)bh * n + i] * 1.44269504088896340736f : 0.f;
    d_s[i] = 0.f;
  }
  const int64_t orow0 = L.o0;
  const int nblk = (n + 15) / 16;
  DqOperands xq;
  if (wave < nblk) dq_load(xq, qbase, tok_stride, o + orow0, d_o + orow0, ostr, wave, lane, n);   // in flight together with the K / V images
  ATTN_STAMP(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ATTN_STAMP(1);
  for (int qb = wave; qb < nblk; qb += nwaves) {   // ------------------------------------ phase 1: dQ, D
    const int q = qb * 16 + (lane & 15);
    DqOperands nx;
    if (qb + nwaves < nblk) dq_load(nx, qbase, tok_stride, o + orow0, d_o + orow0, ostr, qb + nwaves, lane, n);
    const float d = dq_block<NTP>(k_rm, v_rm, xq, tok_stride, dqkv + L.q0, qb, lane, n, scale, lse_s[min(q, n - 1)]);
    if ((lane >> 4) == 0 && q < n) d_s[q] = d;      // stays in LDS for phase 2
    xq = nx;
  }
  DkvOperands xk;
  if (wave < nblk) dkv_load(xk, qbase + inner, tok_stride, inner, wave, lane, n);   // L2 hits, in flight across the barrier and the restaging
  ATTN_STAMP(2);
  __syncthreads();               // every wave is done with the K / V images; D is complete
  ATTN_STAMP(3);
  stage_head_dma(qbase, tok_stride, n, NQP, q_rm, zero_page, wave, lane, nwaves);
  stage_head_dma(d_o + orow0, ostr, n, NQP, do_rm, zero_page, wave, lane, nwaves);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ATTN_STAMP(4);
  for (int kb = wave; kb < nblk; kb += nwaves) {   // ------------------------------------ phase 2: dK, dV
    DkvOperands nx;
    if (kb + nwaves < nblk) dkv_load(nx, qbase + inner, tok_stride, inner, kb + nwaves, lane, n);
    dkv_block<NTP>(q_rm, do_rm, lse_s, d_s, xk, tok_stride, inner, dqkv + L.q0 + inner, kb, lane, n, scale);
    xk = nx;
  }
  ATTN_STAMP(5);
}

template <typename K>
void set_smem(K kern, int bytes) {   // one attribute call per distinct kernel (function-pointer keyed)
  vitx_set_max_smem((const void*)kern, bytes);
}

// 8 waves: 13 one-block waves measured 16 % slower in the backward; 7 waves (tools/probe_attn, r4s) the same forward and a 6 % slower backward
#ifdef VITX_ATTN_PROBE   // tools/probe_attn: workgroup size and a 12-tile instantiation (192 keys: 48 KiB of images, three workgroups per CU) as A/B switches
int vitx_attn_probe_threads = 512;
inline int att_threads(int n) { (void)n; return vitx_attn_probe_threads; }
inline int pick_ntp(int n) { return n <= 64 ? 4 : n <= 96 ? 6 : n <= 192 ? 12 : n <= 224 ? 14 : 18; }
#else
inline int att_threads(int n) { (void)n; return 512; }
inline int pick_ntp(int n) { return n <= 64 ? 4 : n <= 96 ? 6 : n <= 224 ? 14 : 18; }
#endif

}  // namespace

bool attn_bf16_supported(int n, int dim_head) { return dim_head == DH && n >= 1 && n <= 288; }

#define VITX_NTP_DISPATCH(ntp, CALL) \
  do { if ((ntp) == 4) { CALL(4); } else if ((ntp) == 6) { CALL(6); } else if ((ntp) == 14) { CALL(14); } else if ((ntp) == 12) { VITX_NTP12(CALL); } else { CALL(18); } } while (0)
#ifdef VITX_ATTN_PROBE
#define VITX_NTP12(CALL) CALL(12)
#else
#define VITX_NTP12(CALL) CALL(14)   /* (never selected outside the probe build) */
#endif

int vitx_attn_probe_planar = 0;   // tools/probe_attn sets it (layout 1 of AttnLayout); the engine's tensors are layout 0

void launch_attn_bf16_fwd(const bf16_t* qkv, bf16_t* o, float* lse, int b, int n, int h, float scale, const bf16_t* zero_page, int reverse, hipStream_t s) {
  const int ntp = pick_ntp(n);
  const int nkp = 16 * ntp;
  const int smem = 2 * nkp * ROWB;
  // (image, head) tasks from the last to the first: qkv (232 MB at ViT-B/16, written front to back by the GEMM before) is read newest rows
  // first, while they are still in the 256 MB memory-side cache (1.24 -> 1.15 ms per step); `reverse` = bit 4 of the engine's reverse_mask
  // (VITX_REVERSE, read once per handle)
#define CALL(NTP) { set_smem(attn_fwd_kernel<NTP>, smem); hipLaunchKernelGGL(attn_fwd_kernel<NTP>, dim3(b * h), dim3(att_threads(n)), smem, s, qkv, o, lse, n, h, scale, zero_page, reverse, vitx_attn_probe_planar); }
  VITX_NTP_DISPATCH(ntp, CALL);
#undef CALL
}

void launch_attn_bf16_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* d_o, const float* lse, float* dsum_ws, bf16_t* dqkv, int b,
                          int n, int h, float scale, const bf16_t* zero_page, hipStream_t s) {
  const int ntp = pick_ntp(n);
  const int np = 16 * ntp;
  const int smem_dq = 2 * np * ROWB;
  const int smem_dkv = 2 * np * ROWB + 2 * np * 4;
  const char* split_env = vitx_env("VITX_ATTN_BWD_SPLIT");   // A/B and the bit-identity test: the two-launch form (read per call)
  const bool split = split_env && atoi(split_env) != 0;
  if (!split) {
#define CALLF(NTP) { set_smem(attn_bwd_fused_kernel<NTP>, smem_dkv); hipLaunchKernelGGL(attn_bwd_fused_kernel<NTP>, dim3(b * h), dim3(att_threads(n)), smem_dkv, s, qkv, o, d_o, lse, dqkv, n, h, scale, zero_page, vitx_attn_probe_planar); }
    VITX_NTP_DISPATCH(ntp, CALLF);
#undef CALLF
    return;
  }
#define CALL(NTP)                                                                                                                  \
  {                                                                                                                                \
    set_smem(attn_bwd_dq_kernel<NTP>, smem_dq);                                                                                    \
    set_smem(attn_bwd_dkv_kernel<NTP>, smem_dkv);                                                                                  \
    hipLaunchKernelGGL(attn_bwd_dq_kernel<NTP>, dim3(b * h), dim3(att_threads(n)), smem_dq, s, qkv, o, d_o, lse, dsum_ws, dqkv, n, h, scale, zero_page);   \
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<NTP>, dim3(b * h), dim3(att_threads(n)), smem_dkv, s, qkv, d_o, lse, dsum_ws, dqkv, n, h, scale, zero_page);    \
  }
  VITX_NTP_DISPATCH(ntp, CALL);
#undef CALL
}
