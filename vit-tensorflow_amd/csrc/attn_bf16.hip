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

namespace {

using namespace attn_lds;

constexpr int ATT_THREADS = 512;   // 8 waves share one head's LDS image (2 per SIMD)

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
constexpr int QB_BWD = 1;   // backward: QB = 2 costs a workgroup of occupancy (146 / 212 VGPRs) and measured slower

template <int NTP>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                       int n, int h, float scale, const bf16_t* __restrict__ zero_page, int reverse) {
  constexpr int NKP = 16 * NTP, QB = QB_FWD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_rm = smem;                                   // [NKP][128 B] swizzled
  char* v_rm = smem + NKP * ROWB;                      // [NKP][128 B] swizzled (read through the hardware transpose)
  const int bh = reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x, bi = bh / h, hi = bh - bi * h;   // reverse: newest qkv rows first (memory-side cache)
  const int inner = h * DH;
  const int64_t tok_stride = 3 * (int64_t)inner;
  const bf16_t* qbase = qkv + (int64_t)bi * n * tok_stride + hi * DH;
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stage_head_dma(qbase + inner, tok_stride, n, NKP, k_rm, zero_page, wave, lane, nwaves);
  stage_head_dma(qbase + 2 * inner, tok_stride, n, NKP, v_rm, zero_page, wave, lane, nwaves);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int qi = lane & 15, g = lane >> 4;
  const int nqb = (n + 16 * QB - 1) / (16 * QB);
  const float sl2 = scale * 1.44269504088896340736f;
  for (int qb = wave; qb < nqb; qb += nwaves) {
    int q[QB];
    bf16x8 qf[QB][2];
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      q[s] = (qb * QB + s) * 16 + qi;
      const int qc = min(q[s], n - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qf[s][ks] = *(const bf16x8*)(qbase + (int64_t)qc * tok_stride + (g + 4 * ks) * 8);
    }
    // pass 1: row maxima of the scores.  S^T tile t: lane holds S[query qi][key 16t + 4g + r].
    float m[QB];
#pragma unroll
    for (int s = 0; s < QB; ++s) m[s] = -INFINITY;
    // only the last key tiles reach past n: the key mask (a compare + select per score, in kernels whose softmax arithmetic keeps
    // the VALU as busy as the matrix pipe) is applied there alone
    const int t_full = n >> 4;                      // tiles [0, t_full) hold valid keys only
#pragma unroll 2
    for (int t = 0; t < t_full; ++t) {
      const bf16x8 kf0 = frag_rm(k_rm, t * 16 + qi, g), kf1 = frag_rm(k_rm, t * 16 + qi, g + 4);
#pragma unroll
      for (int s = 0; s < QB; ++s) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = mfma16(kf0, qf[s][0], a);
        a = mfma16(kf1, qf[s][1], a);
        m[s] = fmaxf(fmaxf(m[s], fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
      }
    }
    for (int t = t_full; t < NTP && t * 16 < n; ++t) {
      const bf16x8 kf0 = frag_rm(k_rm, t * 16 + qi, g), kf1 = frag_rm(k_rm, t * 16 + qi, g + 4);
#pragma unroll
      for (int s = 0; s < QB; ++s) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = mfma16(kf0, qf[s][0], a);
        a = mfma16(kf1, qf[s][1], a);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[s] = fmaxf(m[s], (t * 16 + 4 * g + r) < n ? a[r] : -INFINITY);
      }
    }
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      m[s] = fmaxf(m[s], __shfl_xor(m[s], 16, 64));
      m[s] = fmaxf(m[s], __shfl_xor(m[s], 32, 64));
      m[s] *= sl2;
    }
    // pass 2: recompute the tile pair, p = 2^(s - m), accumulate the row sum and O^T += V^T P^T (unnormalised)
    float l[QB];
    f32x4 oacc[QB][4];
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      l[s] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) oacc[s][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int u_full = n >> 5;                      // tile pairs [0, u_full) hold valid keys only
    const int u_end = min(NTP / 2, (n + 31) >> 5);  // pairs beyond hold no valid key at all (P = 0: nothing to accumulate)
#pragma unroll 1
    for (int u = 0; u < u_end; ++u) {
      f32x4 p[QB][2];
      const bool masked = u >= u_full;              // wave-uniform
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * u + tt;
        const bf16x8 kf0 = frag_rm(k_rm, t * 16 + qi, g), kf1 = frag_rm(k_rm, t * 16 + qi, g + 4);
#pragma unroll
        for (int s = 0; s < QB; ++s) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          a = mfma16(kf0, qf[s][0], a);
          a = mfma16(kf1, qf[s][1], a);
          if (!masked) {
#pragma unroll
            for (int r = 0; r < 4; ++r) p[s][tt][r] = fast_exp2(fmaf(a[r], sl2, -m[s]));
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) p[s][tt][r] = (t * 16 + 4 * g + r) < n ? fast_exp2(fmaf(a[r], sl2, -m[s])) : 0.f;
          }
          l[s] += (p[s][tt][0] + p[s][tt][1]) + (p[s][tt][2] + p[s][tt][3]);
        }
      }
      bf16x8 pf[QB];
#pragma unroll
      for (int s = 0; s < QB; ++s) pf[s] = pack8(p[s][0], p[s][1]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 vf = frag_trr(v_rm, c, u, lane);
#pragma unroll
        for (int s = 0; s < QB; ++s) oacc[s][c] = mfma16(vf, pf[s], oacc[s][c]);
      }
    }
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      float ls = l[s];
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      const float inv_l = 1.0f / ls;
      if (g == 0 && q[s] < n) lse[(int64_t)bh * n + q[s]] = (m[s] + log2f(ls)) * 0.69314718055994530942f;  // natural-log LSE
      if (q[s] < n) {
        bf16_t* op = o + ((int64_t)bi * n + q[s]) * inner + hi * DH;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          bf16x4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(oacc[s][c][r] * inv_l);
          *(bf16x4*)(op + 16 * c + 4 * g) = ov;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ backward: dQ (+ row sums D)
template <int NTP>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                          const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                          float* __restrict__ dsum, bf16_t* __restrict__ dqkv, int n, int h, float scale,
                                                          const bf16_t* __restrict__ zero_page) {
  constexpr int NKP = 16 * NTP, QB = QB_BWD;
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

  const int qi = lane & 15, g = lane >> 4;
  const int nqb = (n + 16 * QB - 1) / (16 * QB);
  const float sl2 = scale * 1.44269504088896340736f;
  for (int qb = wave; qb < nqb; qb += nwaves) {
    int q[QB];
    bf16x8 qf[QB][2], dof[QB][2];
    float dpart[QB], l2[QB];
    f32x4 dq[QB][4];
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      q[s] = (qb * QB + s) * 16 + qi;
      const int qc = min(q[s], n - 1);
      const int64_t orow = ((int64_t)bi * n + qc) * inner + hi * DH;
      float dp_ = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[s][ks] = *(const bf16x8*)(qbase + (int64_t)qc * tok_stride + (g + 4 * ks) * 8);
        dof[s][ks] = *(const bf16x8*)(d_o + orow + (g + 4 * ks) * 8);
        const bf16x8 of = *(const bf16x8*)(o + orow + (g + 4 * ks) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dp_ += (float)dof[s][ks][e] * (float)of[e];
      }
      dp_ += __shfl_xor(dp_, 16, 64);
      dp_ += __shfl_xor(dp_, 32, 64);   // D[q] = sum_d dO*O
      if (g == 0 && q[s] < n) dsum[(int64_t)bh * n + q[s]] = dp_;
      dpart[s] = dp_;
      l2[s] = lse[(int64_t)bh * n + qc] * 1.44269504088896340736f;
#pragma unroll
      for (int c = 0; c < 4; ++c) dq[s][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // key masks only where a tile pair reaches past n (wave-uniform branch); query rows >= n of the last block compute on the
    // clamped row n-1 (finite) and are never stored, so they need no mask here
    const int u_full = n >> 5, u_end = min(NTP / 2, (n + 31) >> 5);
#pragma unroll 1
    for (int u = 0; u < u_end; ++u) {
      f32x4 ds[QB][2];
      const bool masked = u >= u_full;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * u + tt;
        const bf16x8 kf0 = frag_rm(k_rm, t * 16 + qi, g), kf1 = frag_rm(k_rm, t * 16 + qi, g + 4);
        const bf16x8 vf0 = frag_rm(v_rm, t * 16 + qi, g), vf1 = frag_rm(v_rm, t * 16 + qi, g + 4);
#pragma unroll
        for (int s = 0; s < QB; ++s) {
          f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sa = mfma16(kf0, qf[s][0], sa);
          sa = mfma16(kf1, qf[s][1], sa);
          dp = mfma16(vf0, dof[s][0], dp);
          dp = mfma16(vf1, dof[s][1], dp);
          const float nds = -dpart[s] * scale;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = fast_exp2(fmaf(sa[r], sl2, -l2[s]));
            if (masked) p = (t * 16 + 4 * g + r) < n ? p : 0.f;
            ds[s][tt][r] = p * fmaf(dp[r], scale, nds);
          }
        }
      }
      bf16x8 dsf[QB];
#pragma unroll
      for (int s = 0; s < QB; ++s) dsf[s] = pack8(ds[s][0], ds[s][1]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 kt = frag_trr(k_rm, c, u, lane);
#pragma unroll
        for (int s = 0; s < QB; ++s) dq[s][c] = mfma16(kt, dsf[s], dq[s][c]);
      }
    }
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      if (q[s] < n) {
        bf16_t* dp_out = dqkv + ((int64_t)bi * n + q[s]) * tok_stride + hi * DH;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          bf16x4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)dq[s][c][r];
          *(bf16x4*)(dp_out + 16 * c + 4 * g) = ov;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
template <int NTP>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           bf16_t* __restrict__ dqkv, int n, int h, float scale, const bf16_t* __restrict__ zero_page) {
  constexpr int NQP = 16 * NTP, QB = QB_BWD;
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

  const int ki = lane & 15, g = lane >> 4;
  const int nkb = (n + 16 * QB - 1) / (16 * QB);
  const float sl2 = scale * 1.44269504088896340736f;
  for (int kb = wave; kb < nkb; kb += nwaves) {
    int key[QB];
    bf16x8 kf[QB][2], vf[QB][2];
    f32x4 dk[QB][4], dv[QB][4];
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      key[s] = (kb * QB + s) * 16 + ki;
      const int kc = min(key[s], n - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[s][ks] = *(const bf16x8*)(qbase + inner + (int64_t)kc * tok_stride + (g + 4 * ks) * 8);
        vf[s][ks] = *(const bf16x8*)(qbase + 2 * inner + (int64_t)kc * tok_stride + (g + 4 * ks) * 8);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) { dk[s][c] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[s][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    // No masks: query rows >= n are zero rows of q / dO with lse = D = 0 staged above, so their P = 1 meets dO = 0 and their
    // dS = 1 * (0 - 0); key lanes >= n compute on the clamped key n-1 and are never stored.  Tile pairs past n are skipped.
    const int u_end = min(NTP / 2, (n + 31) >> 5);
#pragma unroll 1
    for (int u = 0; u < u_end; ++u) {
      f32x4 pp[QB][2], ds[QB][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * u + tt;
        const bf16x8 qa0 = frag_rm(q_rm, t * 16 + ki, g), qa1 = frag_rm(q_rm, t * 16 + ki, g + 4);
        const bf16x8 da0 = frag_rm(do_rm, t * 16 + ki, g), da1 = frag_rm(do_rm, t * 16 + ki, g + 4);
        float lq[4], dd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { lq[r] = lse_s[t * 16 + 4 * g + r]; dd[r] = d_s[t * 16 + 4 * g + r]; }
#pragma unroll
        for (int s = 0; s < QB; ++s) {
          f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sa = mfma16(qa0, kf[s][0], sa);     // S[query 16t+4g+r][key ki]
          sa = mfma16(qa1, kf[s][1], sa);
          dp = mfma16(da0, vf[s][0], dp);     // dP same layout
          dp = mfma16(da1, vf[s][1], dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(fmaf(sa[r], sl2, -lq[r]));
            pp[s][tt][r] = p;
            ds[s][tt][r] = p * ((dp[r] - dd[r]) * scale);
          }
        }
      }
      bf16x8 pf[QB], dsf[QB];
#pragma unroll
      for (int s = 0; s < QB; ++s) { pf[s] = pack8(pp[s][0], pp[s][1]); dsf[s] = pack8(ds[s][0], ds[s][1]); }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 a1 = frag_trr(do_rm, c, u, lane), a2 = frag_trr(q_rm, c, u, lane);
#pragma unroll
        for (int s = 0; s < QB; ++s) {
          dv[s][c] = mfma16(a1, pf[s], dv[s][c]);
          dk[s][c] = mfma16(a2, dsf[s], dk[s][c]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      if (key[s] < n) {
        bf16_t* dkp = dqkv + ((int64_t)bi * n + key[s]) * tok_stride + inner + hi * DH;
        bf16_t* dvp = dkp + inner;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          bf16x4 a, b2;
#pragma unroll
          for (int r = 0; r < 4; ++r) { a[r] = (bf16_t)dk[s][c][r]; b2[r] = (bf16_t)dv[s][c][r]; }
          *(bf16x4*)(dkp + 16 * c + 4 * g) = a;
          *(bf16x4*)(dvp + 16 * c + 4 * g) = b2;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ backward, both phases in one launch
// Phase 1 is the dQ kernel above, phase 2 the dK / dV kernel below, run back to back by the SAME workgroup on the same two LDS
// buffers (K, V images, then Q, dO images).  The arithmetic is unchanged (same bits); what changes is the traffic: as two launches
// each pass fetched its head's q, k, v, dO from HBM (310 MB per pass at ViT-B/16, more than the 256-MB MALL holds between them),
// here phase 2 finds them in the L2 its own phase 1 pulled them through 10-20 us earlier, and the row sums D never leave LDS.
template <int NTP>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                             const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dqkv, int n, int h, float scale,
                                                             const bf16_t* __restrict__ zero_page) {
  constexpr int NKP = 16 * NTP, NQP = 16 * NTP, QB = QB_BWD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_rm = smem;                                 // phase 1: K, V      phase 2: Q, dO
  char* v_rm = smem + NKP * ROWB;
  char* q_rm = k_rm;
  char* do_rm = v_rm;
  float* lse_s = (float*)(smem + 2 * NKP * ROWB);    // [NQP] (log2 domain), rows >= n: 0
  float* d_s = lse_s + NQP;                          // [NQP] D[q] = sum_d dO O, rows >= n: 0
  const int bh = blockIdx.x, bi = bh / h, hi = bh - bi * h;
  const int inner = h * DH;
  const int64_t tok_stride = 3 * (int64_t)inner;
  const bf16_t* qbase = qkv + (int64_t)bi * n * tok_stride + hi * DH;
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stage_head_dma(qbase + inner, tok_stride, n, NKP, k_rm, zero_page, wave, lane, nwaves);
  stage_head_dma(qbase + 2 * inner, tok_stride, n, NKP, v_rm, zero_page, wave, lane, nwaves);
  for (int i = tid; i < NQP; i += blockDim.x) {
    lse_s[i] = i < n ? lse[(int64_t)bh * n + i] * 1.44269504088896340736f : 0.f;
    d_s[i] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {   // ---------------------------------------------------------------- phase 1: dQ, D
    const int qi = lane & 15, g = lane >> 4;
    const int nqb = (n + 16 * QB - 1) / (16 * QB);
    const float sl2 = scale * 1.44269504088896340736f;
    for (int qb = wave; qb < nqb; qb += nwaves) {
      int q[QB];
      bf16x8 qf[QB][2], dof[QB][2];
      float dpart[QB], l2[QB];
      f32x4 dq[QB][4];
  #pragma unroll
      for (int s = 0; s < QB; ++s) {
        q[s] = (qb * QB + s) * 16 + qi;
        const int qc = min(q[s], n - 1);
        const int64_t orow = ((int64_t)bi * n + qc) * inner + hi * DH;
        float dp_ = 0.f;
  #pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          qf[s][ks] = *(const bf16x8*)(qbase + (int64_t)qc * tok_stride + (g + 4 * ks) * 8);
          dof[s][ks] = *(const bf16x8*)(d_o + orow + (g + 4 * ks) * 8);
          const bf16x8 of = *(const bf16x8*)(o + orow + (g + 4 * ks) * 8);
  #pragma unroll
          for (int e = 0; e < 8; ++e) dp_ += (float)dof[s][ks][e] * (float)of[e];
        }
        dp_ += __shfl_xor(dp_, 16, 64);
        dp_ += __shfl_xor(dp_, 32, 64);   // D[q] = sum_d dO*O
        if (g == 0 && q[s] < n) d_s[q[s]] = dp_;        // stays in LDS for phase 2
        dpart[s] = dp_;
        l2[s] = lse_s[qc];
  #pragma unroll
        for (int c = 0; c < 4; ++c) dq[s][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // key masks only where a tile pair reaches past n (wave-uniform branch); query rows >= n of the last block compute on the
      // clamped row n-1 (finite) and are never stored, so they need no mask here
      const int u_full = n >> 5, u_end = min(NTP / 2, (n + 31) >> 5);
  #pragma unroll 1
      for (int u = 0; u < u_end; ++u) {
        f32x4 ds[QB][2];
        const bool masked = u >= u_full;
  #pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * u + tt;
          const bf16x8 kf0 = frag_rm(k_rm, t * 16 + qi, g), kf1 = frag_rm(k_rm, t * 16 + qi, g + 4);
          const bf16x8 vf0 = frag_rm(v_rm, t * 16 + qi, g), vf1 = frag_rm(v_rm, t * 16 + qi, g + 4);
  #pragma unroll
          for (int s = 0; s < QB; ++s) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            sa = mfma16(kf0, qf[s][0], sa);
            sa = mfma16(kf1, qf[s][1], sa);
            dp = mfma16(vf0, dof[s][0], dp);
            dp = mfma16(vf1, dof[s][1], dp);
            const float nds = -dpart[s] * scale;
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = fast_exp2(fmaf(sa[r], sl2, -l2[s]));
              if (masked) p = (t * 16 + 4 * g + r) < n ? p : 0.f;
              ds[s][tt][r] = p * fmaf(dp[r], scale, nds);
            }
          }
        }
        bf16x8 dsf[QB];
  #pragma unroll
        for (int s = 0; s < QB; ++s) dsf[s] = pack8(ds[s][0], ds[s][1]);
  #pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bf16x8 kt = frag_trr(k_rm, c, u, lane);
  #pragma unroll
          for (int s = 0; s < QB; ++s) dq[s][c] = mfma16(kt, dsf[s], dq[s][c]);
        }
      }
  #pragma unroll
      for (int s = 0; s < QB; ++s) {
        if (q[s] < n) {
          bf16_t* dp_out = dqkv + ((int64_t)bi * n + q[s]) * tok_stride + hi * DH;
  #pragma unroll
          for (int c = 0; c < 4; ++c) {
            bf16x4 ov;
  #pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)dq[s][c][r];
            *(bf16x4*)(dp_out + 16 * c + 4 * g) = ov;
          }
        }
      }
    }
  }
  __syncthreads();               // every wave is done with the K / V images; D is complete
  stage_head_dma(qbase, tok_stride, n, NQP, q_rm, zero_page, wave, lane, nwaves);
  stage_head_dma(d_o + (int64_t)bi * n * inner + hi * DH, inner, n, NQP, do_rm, zero_page, wave, lane, nwaves);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {   // ---------------------------------------------------------------- phase 2: dK, dV
    const int ki = lane & 15, g = lane >> 4;
    const int nkb = (n + 16 * QB - 1) / (16 * QB);
    const float sl2 = scale * 1.44269504088896340736f;
    for (int kb = wave; kb < nkb; kb += nwaves) {
      int key[QB];
      bf16x8 kf[QB][2], vf[QB][2];
      f32x4 dk[QB][4], dv[QB][4];
  #pragma unroll
      for (int s = 0; s < QB; ++s) {
        key[s] = (kb * QB + s) * 16 + ki;
        const int kc = min(key[s], n - 1);
  #pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          kf[s][ks] = *(const bf16x8*)(qbase + inner + (int64_t)kc * tok_stride + (g + 4 * ks) * 8);
          vf[s][ks] = *(const bf16x8*)(qbase + 2 * inner + (int64_t)kc * tok_stride + (g + 4 * ks) * 8);
        }
  #pragma unroll
        for (int c = 0; c < 4; ++c) { dk[s][c] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[s][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
      // No masks: query rows >= n are zero rows of q / dO with lse = D = 0 staged above, so their P = 1 meets dO = 0 and their
      // dS = 1 * (0 - 0); key lanes >= n compute on the clamped key n-1 and are never stored.  Tile pairs past n are skipped.
      const int u_end = min(NTP / 2, (n + 31) >> 5);
  #pragma unroll 1
      for (int u = 0; u < u_end; ++u) {
        f32x4 pp[QB][2], ds[QB][2];
  #pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * u + tt;
          const bf16x8 qa0 = frag_rm(q_rm, t * 16 + ki, g), qa1 = frag_rm(q_rm, t * 16 + ki, g + 4);
          const bf16x8 da0 = frag_rm(do_rm, t * 16 + ki, g), da1 = frag_rm(do_rm, t * 16 + ki, g + 4);
          float lq[4], dd[4];
  #pragma unroll
          for (int r = 0; r < 4; ++r) { lq[r] = lse_s[t * 16 + 4 * g + r]; dd[r] = d_s[t * 16 + 4 * g + r]; }
  #pragma unroll
          for (int s = 0; s < QB; ++s) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            sa = mfma16(qa0, kf[s][0], sa);     // S[query 16t+4g+r][key ki]
            sa = mfma16(qa1, kf[s][1], sa);
            dp = mfma16(da0, vf[s][0], dp);     // dP same layout
            dp = mfma16(da1, vf[s][1], dp);
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = fast_exp2(fmaf(sa[r], sl2, -lq[r]));
              pp[s][tt][r] = p;
              ds[s][tt][r] = p * ((dp[r] - dd[r]) * scale);
            }
          }
        }
        bf16x8 pf[QB], dsf[QB];
  #pragma unroll
        for (int s = 0; s < QB; ++s) { pf[s] = pack8(pp[s][0], pp[s][1]); dsf[s] = pack8(ds[s][0], ds[s][1]); }
  #pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bf16x8 a1 = frag_trr(do_rm, c, u, lane), a2 = frag_trr(q_rm, c, u, lane);
  #pragma unroll
          for (int s = 0; s < QB; ++s) {
            dv[s][c] = mfma16(a1, pf[s], dv[s][c]);
            dk[s][c] = mfma16(a2, dsf[s], dk[s][c]);
          }
        }
      }
  #pragma unroll
      for (int s = 0; s < QB; ++s) {
        if (key[s] < n) {
          bf16_t* dkp = dqkv + ((int64_t)bi * n + key[s]) * tok_stride + inner + hi * DH;
          bf16_t* dvp = dkp + inner;
  #pragma unroll
          for (int c = 0; c < 4; ++c) {
            bf16x4 a, b2;
  #pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = (bf16_t)dk[s][c][r]; b2[r] = (bf16_t)dv[s][c][r]; }
            *(bf16x4*)(dkp + 16 * c + 4 * g) = a;
            *(bf16x4*)(dvp + 16 * c + 4 * g) = b2;
          }
        }
      }
    }
  }
}

template <typename K>
void set_smem(K kern, int bytes) {   // one attribute call per distinct kernel (function-pointer keyed)
  static const void* done[32];
  static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern) return;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (ndone < 32) done[ndone++] = (const void*)kern;
}

inline int att_threads(int n) { (void)n; return 512; }   // 8 waves measured best (13 one-block waves: bwd 16 % slower)
inline int pick_ntp(int n) { return n <= 64 ? 4 : n <= 96 ? 6 : n <= 224 ? 14 : 18; }

}  // namespace

bool attn_bf16_supported(int n, int dim_head) { return dim_head == DH && n >= 1 && n <= 288; }

#define VITX_NTP_DISPATCH(ntp, CALL) \
  do { if ((ntp) == 4) { CALL(4); } else if ((ntp) == 6) { CALL(6); } else if ((ntp) == 14) { CALL(14); } else { CALL(18); } } while (0)

void launch_attn_bf16_fwd(const bf16_t* qkv, bf16_t* o, float* lse, int b, int n, int h, float scale, const bf16_t* zero_page, int reverse, hipStream_t s) {
  const int ntp = pick_ntp(n);
  const int nkp = 16 * ntp;
  const int smem = 2 * nkp * ROWB;
  // (image, head) tasks from the last to the first: qkv (232 MB at ViT-B/16, written front to back by the GEMM before) is read newest rows
  // first, while they are still in the 256 MB memory-side cache (1.24 -> 1.15 ms per step); `reverse` = bit 4 of the engine's reverse_mask
  // (VITX_REVERSE, read once per handle)
#define CALL(NTP) { set_smem(attn_fwd_kernel<NTP>, smem); hipLaunchKernelGGL(attn_fwd_kernel<NTP>, dim3(b * h), dim3(att_threads(n)), smem, s, qkv, o, lse, n, h, scale, zero_page, reverse); }
  VITX_NTP_DISPATCH(ntp, CALL);
#undef CALL
}

void launch_attn_bf16_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* d_o, const float* lse, float* dsum_ws, bf16_t* dqkv, int b,
                          int n, int h, float scale, const bf16_t* zero_page, hipStream_t s) {
  const int ntp = pick_ntp(n);
  const int np = 16 * ntp;
  const int smem_dq = 2 * np * ROWB;
  const int smem_dkv = 2 * np * ROWB + 2 * np * 4;
  const char* split_env = getenv("VITX_ATTN_BWD_SPLIT");   // A/B and the bit-identity test: the two-launch form (read per call)
  const bool split = split_env && atoi(split_env) != 0;
  if (!split) {
#define CALLF(NTP) { set_smem(attn_bwd_fused_kernel<NTP>, smem_dkv); hipLaunchKernelGGL(attn_bwd_fused_kernel<NTP>, dim3(b * h), dim3(att_threads(n)), smem_dkv, s, qkv, o, d_o, lse, dqkv, n, h, scale, zero_page); }
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
