// DeepViT Re-attention (deepvit.py:73-91) as ONE forward kernel in the bf16 mode:
//   dots = q k^T * scale -> softmax over keys -> attn' = einsum('b h i j, h g -> b g i j', attn, reattn_weights)
//   -> LayerNorm over the head axis at every (b, i, j) -> out = attn' v -> 'b h n d -> b n (h d)'.
// The head mix needs every head of a (query, key) point at once, so a workgroup owns a 16-query tile of one image for ALL heads:
//   stage 1  wave w computes S^T = K Q^T for its heads on the matrix pipe (v_mfma_f32_16x16x32_bf16; K and Q fragments come
//            straight from the packed qkv rows: each is used once per workgroup, there is nothing to stage), masks and scales,
//            does the softmax in registers (a lane owns one query column: in-lane + two xor-shuffles) and leaves P in LDS
//            as fp32 [head][query][key];
//   stage 2  one thread per (query, key) point reads its H probabilities, applies the H x H mix (mixing matrix in SGPRs via
//            wave-uniform scalar loads) and the LayerNorm over heads in registers, and writes the result as bf16 [head][query][key];
//   stage 3  wave w stages V of its head into a private swizzled LDS image (direct-to-LDS loads) and computes
//            O^T = V^T A^T with hardware-transpose reads of V, storing rows of the merged-head output.
// The [b, h, n, n] tensors touch HBM only as the two fp32 tensors the backward consumes (the softmax and the normalised scores;
// the mixed scores in between are recomputed there from the softmax), written once from registers when `keep` is set; the four launches this replaces (batched QK^T GEMM, row statistics, point kernel,
// batched A V GEMM) wrote and re-read them in between.
#include "kernels.h"
#include "attn_lds.h"

namespace {

using namespace attn_lds;

constexpr int DV_THREADS = 512;            // 8 waves
constexpr int DV_NT = 5;                   // 16-key tiles held in registers: nk <= 80
constexpr int DV_PP = 84;                  // fp32 pitch of a P row in LDS (80 keys + 4: 16-B aligned rows)
constexpr int DV_AP = 104;                 // bf16 pitch of a normalised row (96 keys: three 32-key MFMA k-groups, + 8)
constexpr int DV_VROWS = 96;               // rows of a wave's V image (multiple of 32 for the transpose reads)
constexpr int DV_VBYTES = DV_VROWS * ROWB; // 12 KiB per wave

template <int H>
__global__ __launch_bounds__(DV_THREADS) void deepvit_attn_fwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t ldq, int64_t ldk, int64_t ldv,
    int64_t qb, int64_t kb, int64_t vb, bf16_t* __restrict__ o, int64_t ldo, int64_t ob, const float* __restrict__ w,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ p_keep, float* __restrict__ a2_keep, int keep,
    int nq, int nk, int64_t ld, float scale, float eps, const bf16_t* __restrict__ zero_page, int ntile, int cpi, int xp) {
  constexpr int HPW = (H + 7) / 8;                     // heads per wave
  constexpr int R0 = (H * 16 * DV_PP * 4 > 8 * DV_VBYTES) ? H * 16 * DV_PP * 4 : 8 * DV_VBYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* Pl = (float*)smem;                            // [H][16][DV_PP] fp32; dead after stage 2, reused as the V images
  bf16_t* Al = (bf16_t*)(smem + R0);                   // [H][16][DV_AP] bf16
  // A workgroup owns `ntile / cpi` consecutive query tiles of ONE image (cpi = 1 at the BASELINE.json batch: one image per CU, no
  // partial last round of workgroups): its K fragments stay in registers for all of them, the next tile's Q fragments are
  // requested while the current tile is in stage 2, and V is re-staged from an L2 that already holds it.  Workgroup ids go
  // round-robin over the 8 XCDs, each with its own L2; the logical index is made contiguous per XCD so that the chunks of one
  // image (cpi > 1) meet in one L2.
  const int xcd = blockIdx.x & 7, per = gridDim.x >> 3, rem = gridDim.x & 7;
  const int logical = xcd * per + min(xcd, rem) + (int)(blockIdx.x >> 3);
  const int bi = logical / cpi, chunk = logical - bi * cpi;
  const int t_begin = chunk * ntile / cpi, t_end = (chunk + 1) * ntile / cpi;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;
  const int nt = (nk + 15) >> 4, nu = (nk + 31) >> 5;
  const int64_t plane = (int64_t)nq * ld;

  // key columns nk .. 32 nu - 1 of the normalised scores multiply zero rows of V: zero them once (stage 2 never touches them)
  for (int idx = tid; idx < H * 16 * (DV_AP / 4); idx += DV_THREADS)
    *(bf16x4*)(Al + idx * 4) = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};

  // K fragments of this wave's heads: S^T = K Q^T takes K as the A operand, lane (key qi of tile t, 16-B chunk g)
  bf16x8 kf[HPW][DV_NT][2], qf[HPW][2];
#pragma unroll
  for (int s = 0; s < HPW; ++s) {
    const int head = min(wave * HPW + s, H - 1);
#pragma unroll
    for (int t = 0; t < DV_NT; ++t) {
      const int krow = min(16 * t + qi, nk - 1);
      const bf16_t* kp = k + (int64_t)bi * kb + (int64_t)krow * ldk + head * DH;
      if (t < nt) {
        kf[s][t][0] = *(const bf16x8*)(kp + g * 8);
        kf[s][t][1] = *(const bf16x8*)(kp + (g + 4) * 8);
      } else {
        kf[s][t][0] = zero8(); kf[s][t][1] = zero8();
      }
    }
  }
  auto load_q = [&](int tile, bf16x8 (&dst)[HPW][2]) {
    const int qrow = min(tile * 16 + qi, nq - 1);
#pragma unroll
    for (int s = 0; s < HPW; ++s) {
      const int head = min(wave * HPW + s, H - 1);
      const bf16_t* qp = q + (int64_t)bi * qb + (int64_t)qrow * ldq + head * DH;
      dst[s][0] = *(const bf16x8*)(qp + g * 8);
      dst[s][1] = *(const bf16x8*)(qp + (g + 4) * 8);
    }
  };
  load_q(t_begin, qf);

  // stage-2 roles (see the backward kernel): key slot jl, head quad hq; the mixing matrix as the A operand of v_mfma_f32_16x16x4_f32
  const int jl = lane & 15, hq = lane >> 4;
  float WA[4], gm[4], bt[4];
  bool hv[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int hh = 4 * hq + st;
    hv[st] = hh < H;
    WA[st] = (hh < H && jl < H) ? w[hh * H + jl] : 0.f;
    gm[st] = hh < H ? gamma[hh] : 0.f;
    bt[st] = hh < H ? beta[hh] : 0.f;
  }
  auto qsum = [](float x) {        // sum over the four head quads of a key (lanes l, l ^ 16, l ^ 32, l ^ 48)
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float t = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    const unsigned v2 = __builtin_bit_cast(unsigned, t);
    const auto s2 = __builtin_amdgcn_permlane32_swap(v2, v2, false, false);
    return __builtin_bit_cast(float, (unsigned)s2[0]) + __builtin_bit_cast(float, (unsigned)s2[1]);
  };
  // keep == 2 (round 5): the normalised scores are kept as bf16 [H][nq][ld2], ld2 = nk rounded up to 8 -- their only reader is the dV product, whose
  // loader rounds them to bf16 anyway (same bits, half the bytes written here and read there)
  const bool a2_lp = keep == 2;
  const int ld2 = (nk + 7) & ~7;
  const int64_t plane2 = (int64_t)nq * ld2;
  const __amdgpu_buffer_rsrc_t rsA2 = a2_lp ? __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)a2_keep + (int64_t)bi * H * plane2), 0, (int)(H * plane2 * 2), 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc((void*)(a2_keep + (int64_t)bi * H * plane), 0, (int)(H * plane * 4), 0x00020000);
  const int plane4 = a2_lp ? (int)plane2 * 2 : (int)plane * 4;

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int q0 = tile * 16;
    // ---------------------------------------------------------------- stage 1: S^T, softmax (deepvit.py:79-80)
#pragma unroll
    for (int s = 0; s < HPW; ++s) {
      const int head = wave * HPW + s;
      if (head < H) {
        f32x4 sv[DV_NT];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < DV_NT; ++t) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          if (t < nt) {
            a = mfma16(kf[s][t][0], qf[s][0], a);       // lane: S[query qi][key 16t + 4g + r]
            a = mfma16(kf[s][t][1], qf[s][1], a);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = (16 * t + 4 * g + r) < nk ? a[r] * scale : -INFINITY;
            sv[t][r] = x;
            m = fmaxf(m, x);
          }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < DV_NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pe = fast_exp2((sv[t][r] - m) * 1.44269504088896340736f);   // masked keys: exp2(-inf) = 0
            sv[t][r] = pe;
            l += pe;
          }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const bool st = keep && (q0 + qi) < nq;
        float* prow = p_keep + ((int64_t)bi * H + head) * plane + (int64_t)(q0 + qi) * ld;
#pragma unroll
        for (int t = 0; t < DV_NT; ++t) {
          if (t < nt) {
            const float4 f = make_float4(sv[t][0] * inv, sv[t][1] * inv, sv[t][2] * inv, sv[t][3] * inv);
            *(float4*)(Pl + (head * 16 + qi) * DV_PP + 16 * t + 4 * g) = f;
            if (st && (16 * t + 4 * g) < ld) *(float4*)(prow + 16 * t + 4 * g) = f;   // columns nk..ld-1 get their zeros
          }
        }
      }
    }
    if (tile + 1 < t_end) load_q(tile + 1, qf);         // in flight during stages 2 and 3
    __syncthreads();

    // -------------------------------------------------------------- stage 2: re-attention mix + LayerNorm over heads (deepvit.py:83-84)
    // Round 5: the H x H mix runs on the fp32 matrix pipe with the mixing matrix in registers (lane = (key slot jl, head quad hq), a lane owns heads
    // 4 hq + r of its key; see the backward kernel below for the operand roles) -- wave w takes queries w and w + 8 in groups of 16 keys.  The
    // one-thread-per-point form it replaces read W through scalar loads (2 x H row round trips per point batch: 5 us per tile).
    if (!(xp & 2)) {
#pragma unroll 1
      for (int rr = 0; rr < 2; ++rr) {
        const int i = wave + 8 * rr;
        const bool rv = (q0 + i) < nq;
        if (!rv) continue;   // (wave-uniform) a query row past the end -- 15 of the last tile's 16 at 65 tokens: its column of O^T is never stored
#pragma unroll
        for (int gk = 0; gk < DV_NT; ++gk) {
          if (gk < nt) {
            const int j = 16 * gk + jl;
            const bool valid = rv && j < nk;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float t = Pl[(min(4 * hq + r, H - 1) * 16 + i) * DV_PP + j]; y[r] = (valid && hv[r]) ? t : 0.f; }
            f32x4 vv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; ++st) vv = __builtin_amdgcn_mfma_f32_16x16x4f32(WA[st], y[st], vv, 0, 0, 0);   // mixed scores of heads 4 hq + r
            const float mu = qsum((vv[0] + vv[1]) + (vv[2] + vv[3])) * (1.0f / (float)H);   // (rows >= H of the mix are exact zeros)
            float xh[4], var = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { xh[r] = hv[r] ? vv[r] - mu : 0.f; var += xh[r] * xh[r]; }
            const float rs = rsqrtf(qsum(var) * (1.0f / (float)H) + eps);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int hh = 4 * hq + r;
              const float a2 = xh[r] * rs * gm[r] + bt[r];
              if (hv[r]) Al[(hh * 16 + i) * DV_AP + j] = valid ? (bf16_t)a2 : (bf16_t)0.f;   // (keys nk .. 16 nt - 1 multiply zero rows of V)
              // kept for the backward: valid points only; everything else is sent out of the descriptor's range (dropped)
              if (a2_lp)
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)a2), rsA2,
                                                      (valid && hv[r]) ? (4 * hq * (int)plane2 + (q0 + i) * ld2 + j) * 2 : 0x7ffffff0, r * plane4, 0);
              else
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a2), rsA2,
                                                      (keep && valid && hv[r]) ? (4 * hq * (int)plane + (q0 + i) * (int)ld + j) * 4 : 0x7ffffff0, r * plane4, 0);
            }
          }
        }
      }
    }
    __syncthreads();

    // -------------------------------------------------------------- stage 3: out = attn' v, merged heads (deepvit.py:87-88)
    {
      char* vbuf = smem + wave * DV_VBYTES;
#pragma unroll
      for (int s = 0; s < HPW; ++s) {
        const int head = wave * HPW + s;
        if (head >= H || (xp & 4)) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous head's transpose reads have returned
        stage_head_dma(v + (int64_t)bi * vb + head * DH, ldv, nk, 32 * nu, vbuf, zero_page, 0, lane, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 oacc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int u = 0; u < nu; ++u) {
          // B operand: k-slot (g, e) <-> key 32u + 4g + e (e < 4), 32u + 16 + 4g + (e - 4): the permutation frag_trr applies to V
          const bf16_t* ar = Al + (head * 16 + qi) * DV_AP + 32 * u + 4 * g;
          const bf16x4 lo = *(const bf16x4*)ar, hi = *(const bf16x4*)(ar + 16);
          bf16x8 pf;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) { pf[e2] = lo[e2]; pf[4 + e2] = hi[e2]; }
#pragma unroll
          for (int c = 0; c < 4; ++c) oacc[c] = mfma16(frag_trr(vbuf, c, u, lane), pf, oacc[c]);
        }
        if (q0 + qi < nq) {
          bf16_t* op = o + (int64_t)bi * ob + (int64_t)(q0 + qi) * ldo + head * DH;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)oacc[c][r];
            *(bf16x4*)(op + 16 * c + 4 * g) = ov;
          }
        }
      }
    }
    __syncthreads();   // the V images overlay the P buffer of the next tile
  }
}


// ------------------------------------------------------------------------------------------------ backward (round 5)
// VJP of the chain above up to d(q), with the softmax P the forward kept (deepvit.py:79-88):
//   d(attn') = d(out) v^T -> LayerNorm-over-heads VJP -> re-attention mix VJP (+ dW, dgamma, dbeta) -> softmax VJP = d(dots) -> dq = scale d(dots) k.
// Same ownership as the forward: a workgroup takes 16-query tiles of ONE image for ALL heads.
//   stage 1  wave w computes d(attn')^T = V dO^T for its heads on the matrix pipe (V and dO fragments straight from the packed rows) and leaves it in
//            LDS as fp32 [head][query][key];
//   stage 2  a WAVE per query row, in groups of 16 keys x 4 head quads (see the lane roles below): P of the point's H
//            heads from HBM (what the forward kept), the mixed scores recomputed from it (the forward's FMA order), LayerNorm VJP and mix VJP in
//            registers with the mixing matrix read from LDS (broadcast reads: as scalar loads from memory each of its 2 x H rows cost a ~200-cycle
//            round trip per point batch); the softmax VJP d(dots) = P (d(P) - sum_j d(P) P) follows in the same registers, its row sum being a
//            wave reduction (+ the tail pass's share through LDS), so d(P) never leaves them; d(dots) goes out as fp32 to HBM (the d(k) product
//            reads it) and as bf16 into LDS for stage 3.  The H x H mixing-matrix gradient is summed on the fp32 matrix pipe
//            (v_mfma_f32_16x16x4_f32 over 16-point batches transposed through a per-wave LDS scratch), dgamma / dbeta in lanes;
//   stage 3  wave w stages K of its head into a private swizzled LDS image and computes dq^T = K^T d(dots)^T with hardware-transpose reads of K.
// d(k) = scale d(dots)^T q and d(v) = attn'^T d(out) accumulate over ALL query tiles of an image (80 x 64 fp32 per head and product: 160 registers per
// head on top of everything else) and stay one batched-GEMM pair behind this kernel (engine.hip).  Replaces per block: the d(attn') GEMM, the point
// kernel, two partial reductions, the softmax row kernel and the dq GEMM with their 77-MB [b, h, n, n] round trips in between.
// LDS: R0 = d(attn') (stages 1-2) / the K images (stage 3); R1 = 16 row slots of d(dots) (bf16, [query][head][key]) -- wave w owns the slots of queries
// w and w + 8 and uses the second one as its transpose scratch until that row's d(dots) is written; EX = mixing matrix, row sums of the tail pass.
constexpr int DV_HP = 20;                  // pitch (floats) of a point's row in the transpose scratch: 16 heads + 4
constexpr int DV_SCR = 2 * 16 * DV_HP * 4; // per wave: P and d(mixed) of 16 points (2560 B)
constexpr int DVB_THREADS = 512;
constexpr int DVB_WAVES = 8;

template <int H> constexpr int dv_bwd_rowsz() { return (H * DV_AP * 2 > DV_SCR) ? H * DV_AP * 2 : DV_SCR; }
template <int H> constexpr int dv_bwd_r0() { return (H * 16 * DV_PP * 4 > DVB_WAVES * DV_VBYTES) ? H * 16 * DV_PP * 4 : DVB_WAVES * DV_VBYTES; }

template <int H>
__global__ __launch_bounds__(DVB_THREADS) void deepvit_attn_bwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t ldq, int64_t ldk, int64_t ldv, int64_t qb,
    int64_t kb, int64_t vb, const bf16_t* __restrict__ d_o, int64_t ldo, int64_t ob, const float* __restrict__ p_keep, const float* __restrict__ w,
    const float* __restrict__ gamma, float* __restrict__ ds_out, bf16_t* __restrict__ dq, int64_t lddq, int64_t dqb, float* __restrict__ partial,
    int nq, int nk, int64_t ld, float scale, float eps, const bf16_t* __restrict__ zero_page, int ntile, int cpi, int xp, int ds_bf16) {
  constexpr int HPW = (H + DVB_WAVES - 1) / DVB_WAVES;   // heads per wave (stages 1, 3)
  constexpr int RPW = 16 / DVB_WAVES;                    // query rows per wave (stage 2)
  constexpr int R0 = dv_bwd_r0<H>();
  constexpr int ROWSZ = dv_bwd_rowsz<H>(), R1 = 16 * ROWSZ;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* Dl = (float*)smem;                            // [H][16][DV_PP] fp32 d(attn'); tail points keep their d(P) here; dead after stage 2, then the K images
  char* const r1 = smem + R0;
  // d(dots) of query qi_: [head][DV_AP] bf16.  Wave w owns the slots of queries w, w + 8 (contiguous) and uses the LAST of them as its
  // transpose scratch until that row's d(dots) is written
  auto sl_row = [&](int qi_) { return (bf16_t*)(r1 + ((qi_ & (DVB_WAVES - 1)) * RPW + (qi_ / DVB_WAVES)) * ROWSZ); };
  const int xcd = blockIdx.x & 7, per = gridDim.x >> 3, rem = gridDim.x & 7;
  const int logical = xcd * per + min(xcd, rem) + (int)(blockIdx.x >> 3);
  const int bi = logical / cpi, chunk = logical - bi * cpi;
  const int t_begin = chunk * ntile / cpi, t_end = (chunk + 1) * ntile / cpi;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;
  const int nt = (nk + 15) >> 4, nu = (nk + 31) >> 5;
  const int64_t plane = (int64_t)nq * ld;
  float* xs = (float*)(r1 + (wave * RPW + RPW - 1) * ROWSZ);   // this wave's transpose scratch: [16 points][DV_HP] of P ...
  float* ys = xs + 16 * DV_HP;                                 // ... and of d(mixed)

  // ---- stage-2 roles: lane = (key slot jl = lane & 15, head quad hq = lane >> 4); a lane owns heads 4 hq + r (r < 4) of its key.
  // Both H x H mixes run on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, D[m][n] += sum_k A[m][k] B[k][n] with n = the 16 keys of a group,
  // k <-> the head the lane holds in register st): the mixing matrix sits in 8 registers per lane as the A operands
  //   mix  (deepvit.py:83)  mixed[key][g] = sum_h P[key][h] W[h][g]:      A[m = g][k] = W[4 hq + st][g = jl],   B[k][n] = P of head 4 hq + st
  //   VJP                   d(P)[key][h]  = sum_g d(mixed)[key][g] W[h][g]: A[m = h][k] = W[h = jl][4 hq + st],   B[k][n] = d(mixed) of head 4 hq + st
  // and the results land in the same ownership (D: lane (n = jl, rows 4 hq + r)).  As VALU work the two mixes were 512 FMAs per point with the
  // matrix re-read for every point batch (s_load round trips, or H-element register arrays that spilled by the hundred); the fp32 MFMA has the
  // VALU's FMA rate, takes the 16 x 16 matrix from registers and leaves the VALU to the LayerNorm / softmax arithmetic beside it.
  const int jl = lane & 15, hq = lane >> 4;
  float WA[4], WB[4], gm[4];
  bool hv[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int hh = 4 * hq + st;
    hv[st] = hh < H;
    WA[st] = (hh < H && jl < H) ? w[hh * H + jl] : 0.f;
    WB[st] = (hh < H && jl < H) ? w[jl * H + hh] : 0.f;
    gm[st] = hh < H ? gamma[hh] : 0.f;
  }
  auto qsum = [](float x) {        // sum over the four head quads of a key (lanes l, l ^ 16, l ^ 32, l ^ 48): two half-swaps + adds, no LDS
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float t = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    const unsigned v2 = __builtin_bit_cast(unsigned, t);
    const auto s2 = __builtin_amdgcn_permlane32_swap(v2, v2, false, false);
    return __builtin_bit_cast(float, (unsigned)s2[0]) + __builtin_bit_cast(float, (unsigned)s2[1]);
  };
  auto row16_sum = [](float v2) {  // sum over the 16 key slots of a quad (one DPP row): every lane of the row gets the total
#define DVB_DPP(ctrl) v2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v2), (ctrl), 0xF, 0xF, false))
    DVB_DPP(0xB1); DVB_DPP(0x4E); DVB_DPP(0x124); DVB_DPP(0x128);
#undef DVB_DPP
    return v2;
  };

  f32x4 wacc = {0.f, 0.f, 0.f, 0.f};                   // this wave's share of dW[h][g] (rows 4 (lane >> 4) + r, column lane & 15)
  float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};   // dgamma / dbeta of heads 4 hq + r, this lane's keys

  // V fragments of this wave's heads stay in registers for every query tile (d(attn')^T = V dO^T takes V as the A operand: lane (key qi of tile t,
  // 16-B chunk g)); the next tile's dO fragments are requested while the current tile is in stages 2 and 3 (as the forward does with K and Q)
  bf16x8 vf[HPW][DV_NT][2], dof[HPW][2];
#pragma unroll
  for (int s = 0; s < HPW; ++s) {
    const int head = min(wave * HPW + s, H - 1);
#pragma unroll
    for (int t = 0; t < DV_NT; ++t) {
      const int krow = min(16 * t + qi, nk - 1);
      const bf16_t* vp = v + (int64_t)bi * vb + (int64_t)krow * ldv + head * DH;
      vf[s][t][0] = *(const bf16x8*)(vp + g * 8);
      vf[s][t][1] = *(const bf16x8*)(vp + (g + 4) * 8);
    }
  }
  auto load_do = [&](int tile_) {
    const int qrow = min(tile_ * 16 + qi, nq - 1);
#pragma unroll
    for (int s = 0; s < HPW; ++s) {
      const int head = min(wave * HPW + s, H - 1);
      const bf16_t* dp_ = d_o + (int64_t)bi * ob + (int64_t)qrow * ldo + head * DH;
      dof[s][0] = *(const bf16x8*)(dp_ + g * 8);
      dof[s][1] = *(const bf16x8*)(dp_ + (g + 4) * 8);
    }
  };
  // P of one query row for this lane: [group][head r].  Buffer loads through a descriptor over this image's [H][nq][ld] block: ONE lane offset per
  // row (the group is the instruction's immediate offset, the head a scalar offset), no per-load address registers, and nothing conditional -- a load
  // under a per-lane condition becomes a branch around it with a full `s_waitcnt vmcnt(0)` behind each (twenty serialised memory round trips per
  // row in the first version); reads past the block return zero, the rest is masked afterwards
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p_keep + (int64_t)bi * H * plane), 0, (int)(H * plane * 4), 0x00020000);
  // ds_bf16 (round 5): d(dots) leaves as bf16 [H][nq][ld2], ld2 = nk rounded up to 8 -- its only reader is the dK product, whose loader rounds it anyway
  const int ld2 = (nk + 7) & ~7;
  const int64_t plane2 = (int64_t)nq * ld2;
  const __amdgpu_buffer_rsrc_t rsD = ds_bf16 ? __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)ds_out + (int64_t)bi * H * plane2), 0, (int)(H * plane2 * 2), 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc((void*)(ds_out + (int64_t)bi * H * plane), 0, (int)(H * plane * 4), 0x00020000);
  const int plane4 = (int)plane * 4, planeD = ds_bf16 ? (int)plane2 * 2 : (int)plane * 4;
  auto load_p = [&](int tile_, int rr_, float (&dst)[DV_NT][4]) {
    const int i_ = wave + DVB_WAVES * rr_, qrow = tile_ * 16 + i_;
    const bool rv_ = tile_ < t_end && qrow < nq;
    const int voff = (4 * hq * (int)plane + min(qrow, nq - 1) * (int)ld + jl) * 4;
#pragma unroll
    for (int gk = 0; gk < DV_NT; ++gk) {
      const int j = 16 * gk + jl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsP, voff + 64 * gk, r * plane4, 0));
        dst[gk][r] = (rv_ && j < nk && hv[r]) ? t : 0.f;
      }
    }
  };
  load_do(t_begin);
  float yn[DV_NT][4];
  load_p(t_begin, 0, yn);

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int q0 = tile * 16;
    // ---------------------------------------------------------------- stage 1: d(attn')^T = V dO^T (VJP of deepvit.py:87)
    if (!(xp & 1)) {
#pragma unroll
      for (int s = 0; s < HPW; ++s) {
        const int head = wave * HPW + s;
        if (head < H) {
#pragma unroll
          for (int t = 0; t < DV_NT; ++t) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = mfma16(vf[s][t][0], dof[s][0], a);        // lane: d(attn')[query qi][key 16t + 4g + r]
            a = mfma16(vf[s][t][1], dof[s][1], a);
            *(float4*)(Dl + (head * 16 + qi) * DV_PP + 16 * t + 4 * g) = make_float4(a[0], a[1], a[2], a[3]);
          }
        }
      }
    }
    __syncthreads();

    // ---------------------------------------------------------------- stage 2: LayerNorm-over-heads VJP, mix VJP, softmax VJP (deepvit.py:80-84)
    // wave w takes queries w and w + 8; a query row = nt groups of 16 keys; d(P) and P of the whole row stay in registers until its row sums are complete
    if (!(xp & 2)) {
#pragma unroll 1
      for (int rr = 0; rr < RPW; ++rr) {
        const int i = wave + DVB_WAVES * rr;
        const bool rv = (q0 + i) < nq;
        float y[DV_NT][4], dp[DV_NT][4];
        float rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < DV_NT; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) y[kb][r] = yn[kb][r];
        if (rr + 1 < RPW) load_p(tile, rr + 1, yn); else load_p(tile + 1, 0, yn);   // the next row's P, in flight under this row's arithmetic
        if (!rv) continue;   // (wave-uniform) a query row past the end: contributes nothing to dW / dgamma / dbeta, and its column of dq^T is never stored
#pragma unroll
        for (int kb = 0; kb < DV_NT; ++kb) {
          if (kb < nt) {
            const int j = 16 * kb + jl;
            const bool valid = rv && j < nk;
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float t = Dl[(min(4 * hq + r, H - 1) * 16 + i) * DV_PP + j]; d[r] = (valid && hv[r]) ? t : 0.f; }
            f32x4 vv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; ++st) vv = __builtin_amdgcn_mfma_f32_16x16x4f32(WA[st], y[kb][st], vv, 0, 0, 0);   // mixed scores of heads 4 hq + r
            const float mu = qsum((vv[0] + vv[1]) + (vv[2] + vv[3])) * (1.0f / (float)H);   // (rows >= H of the mix are exact zeros)
            float xh[4], var = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { xh[r] = hv[r] ? vv[r] - mu : 0.f; var += xh[r] * xh[r]; }
            const float rs = rsqrtf(qsum(var) * (1.0f / (float)H) + eps);
            float dm[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              xh[r] *= rs;
              ab[r] += d[r];
              ag[r] += d[r] * xh[r];
              dm[r] = d[r] * gm[r];
              s1 += dm[r];
              s2 += dm[r] * xh[r];
            }
            s1 = qsum(s1) * (1.0f / (float)H);
            s2 = qsum(s2) * (1.0f / (float)H);
#pragma unroll
            for (int r = 0; r < 4; ++r) dm[r] = (valid && hv[r]) ? rs * (dm[r] - s1 - xh[r] * s2) : 0.f;
            // dW[h][g] += sum over the group's 16 keys of P[h] d(mixed)[g]: transposed through the scratch ([key][head] rows), contraction over keys
            if (!(xp & 32)) {
            *(float4*)(xs + jl * DV_HP + 4 * hq) = make_float4(y[kb][0], y[kb][1], y[kb][2], y[kb][3]);
            *(float4*)(ys + jl * DV_HP + 4 * hq) = make_float4(dm[0], dm[1], dm[2], dm[3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float ta[4], tb[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) { ta[st] = xs[(4 * st + hq) * DV_HP + jl]; tb[st] = ys[(4 * st + hq) * DV_HP + jl]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int st = 0; st < 4; ++st) wacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[st], tb[st], wacc, 0, 0, 0);
            }
            f32x4 dpv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; ++st) dpv = __builtin_amdgcn_mfma_f32_16x16x4f32(WB[st], dm[st], dpv, 0, 0, 0);    // d(P) of heads 4 hq + r
#pragma unroll
            for (int r = 0; r < 4; ++r) { dp[kb][r] = dpv[r]; rsum[r] += row16_sum(y[kb][r] * dpv[r]); }
          }
        }
        // softmax VJP (deepvit.py:80): d(dots) = P (d(P) - sum_j d(P) P); fp32 to HBM (the d(k) product reads it), bf16 into this query's LDS slot
        bf16_t* sl = sl_row(i);
#pragma unroll
        for (int kb = 0; kb < DV_NT; ++kb) {
          if (kb < nt) {
            const int j = 16 * kb + jl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int hh = 4 * hq + r;
              const float ds = y[kb][r] * (dp[kb][r] - rsum[r]);      // 0 for masked keys / rows / heads (P = 0)
              if (hv[r]) sl[hh * DV_AP + j] = (bf16_t)ds;
              // columns nk .. ld - 1 get their zeros; everything else (padding rows / columns / heads) is sent out of the descriptor's range: dropped
              if (ds_bf16)
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)ds), rsD,
                                                      (rv && j < ld2 && hv[r] && !(xp & 16)) ? (4 * hq * (int)plane2 + (q0 + i) * ld2 + j) * 2 : 0x7ffffff0, r * planeD, 0);
              else
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ds), rsD,
                                                      (rv && j < ld && hv[r] && !(xp & 16)) ? (4 * hq * (int)plane + (q0 + i) * (int)ld + j) * 4 : 0x7ffffff0, r * planeD, 0);
            }
          }
        }
        if (16 * nt < 32 * nu) {                                      // keys 16 nt .. 32 nu - 1 multiply the zero rows of the K image: zeros
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (hv[r]) sl[(4 * hq + r) * DV_AP + 16 * nt + jl] = (bf16_t)0.f;
        }
      }
    }
    if (tile + 1 < t_end) load_do(tile + 1);              // in flight under stage 3
    __syncthreads();

    // ---------------------------------------------------------------- stage 3: dq = scale d(dots) k (VJP of deepvit.py:79), merged heads
    {
      char* kbuf = smem + wave * DV_VBYTES;
      const bf16_t* slq = sl_row(qi);
#pragma unroll 1
      for (int s = 0; s < HPW; ++s) {
        const int head = wave * HPW + s;
        if (head >= H || (xp & 8)) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous head's transpose reads have returned
        stage_head_dma(k + (int64_t)bi * kb + head * DH, ldk, nk, 32 * nu, kbuf, zero_page, 0, lane, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 oacc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int u = 0; u < nu; ++u) {
          const bf16_t* ar = slq + head * DV_AP + 32 * u + 4 * g;
          const bf16x4 lo = *(const bf16x4*)ar, hi = *(const bf16x4*)(ar + 16);
          bf16x8 pf;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) { pf[e2] = lo[e2]; pf[4 + e2] = hi[e2]; }
#pragma unroll
          for (int c = 0; c < 4; ++c) oacc[c] = mfma16(frag_trr(kbuf, c, u, lane), pf, oacc[c]);
        }
        if (q0 + qi < nq) {
          bf16_t* op = dq + (int64_t)bi * dqb + (int64_t)(q0 + qi) * lddq + head * DH;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(oacc[c][r] * scale);
            *(bf16x4*)(op + 16 * c + 4 * g) = ov;
          }
        }
      }
    }
    __syncthreads();   // the K images overlay d(attn') of the next tile, the scratch overlays d(dots)
  }

  // per-wave partials: [dW (H*H) | dgamma (H) | dbeta (H)], summed in a fixed order by the reduction behind the launch
  float* pw = partial + ((int64_t)blockIdx.x * DVB_WAVES + wave) * (H * H + 2 * H);
  const int g16 = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int hh = 4 * (lane >> 4) + r;
    if (hh < H && g16 < H) pw[hh * H + g16] = wacc[r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {     // dgamma / dbeta of head 4 hq + r: sum over the quad's 16 key slots
    const float sg = row16_sum(ag[r]), sb = row16_sum(ab[r]);
    const int hh = 4 * hq + r;
    if (jl == 0 && hh < H) { pw[H * H + hh] = sg; pw[H * H + H + hh] = sb; }
  }
}

template <int H>
constexpr int dv_bwd_smem() { return dv_bwd_r0<H>() + 16 * dv_bwd_rowsz<H>(); }

template <int H>
constexpr int dv_fwd_smem() {
  return ((H * 16 * DV_PP * 4 > 8 * DV_VBYTES) ? H * 16 * DV_PP * 4 : 8 * DV_VBYTES) + H * 16 * DV_AP * 2;
}

template <typename K>
void dv_set_smem(K kern, int bytes) {
  vitx_set_max_smem((const void*)kern, bytes);
}

}  // namespace

bool deepvit_attn_fused_supported(int h, int dim_head, int nq, int nk) {
  return dim_head == DH && (h == 4 || h == 8 || h == 12 || h == 16) && nq >= 1 && nk >= 1 && nk <= 16 * DV_NT;
}

void launch_deepvit_attn_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv, int64_t qb,
                             int64_t kb, int64_t vb, bf16_t* o, int64_t ldo, int64_t ob, const float* w, const float* gamma,
                             const float* beta, float* p_keep, float* a2_keep, int keep, int b, int h, int nq,
                             int nk, int64_t ld, float scale, float eps, const bf16_t* zero_page, hipStream_t s) {
  const int ntile = (nq + 15) / 16;
  // chunks per image: one workgroup per image once the batch alone fills the chip, otherwise enough chunks to cover the 256 CUs
  int cpi = b >= 192 ? 1 : std::max(1, std::min(ntile, (256 + b - 1) / b));
  if (const char* e = vitx_env("VITX_DV_CPI")) cpi = std::max(1, std::min(ntile, atoi(e)));   // tests: the multi-tile loop at small batches
  static const int xp = [] { const char* e = vitx_env("VITX_DV_XP"); return e ? atoi(e) : 0; }();   // timing experiments: 1 = no kept tensors, 2 = no stage 2, 4 = no stage 3 (WRONG results)
  if (xp & 1) keep = 0;
#define CALL(HT)                                                                                                                  \
  {                                                                                                                               \
    dv_set_smem(deepvit_attn_fwd_kernel<HT>, dv_fwd_smem<HT>());                                                                  \
    hipLaunchKernelGGL(deepvit_attn_fwd_kernel<HT>, dim3(b * cpi), dim3(DV_THREADS), dv_fwd_smem<HT>(), s, q, k, v, ldq, ldk,   \
                       ldv, qb, kb, vb, o, ldo, ob, w, gamma, beta, p_keep, a2_keep, keep, nq, nk, ld, scale, eps,   \
                       zero_page, ntile, cpi, xp);                                                                                \
  }
  if (h == 4) CALL(4) else if (h == 8) CALL(8) else if (h == 12) CALL(12) else CALL(16)
#undef CALL
}

// ---- backward launcher: d(dots) [b, h, nq, ld] fp32 -> ds_out, dq (bf16) in place of the d(q) product, dW / dgamma / dbeta through per-wave partials
static int dv_bwd_cpi(int b, int ntile) { return b >= 192 ? 1 : std::max(1, std::min(ntile, (256 + b - 1) / b)); }
int64_t deepvit_attn_bwd_ws_elems(int b, int h, int nq) {
  const int ntile = (nq + 15) / 16;
  return ((int64_t)b * dv_bwd_cpi(b, ntile) * DVB_WAVES + 40) * ((int64_t)h * h + 2 * h);
}
void launch_deepvit_attn_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv, int64_t qb, int64_t kb, int64_t vb,
                             const bf16_t* d_o, int64_t ldo, int64_t ob, const float* p_keep, const float* w, const float* gamma, float* ds_out,
                             bf16_t* dq, int64_t lddq, int64_t dqb, float* ws, float* dw, float* dgamma, float* dbeta, int b, int h, int nq, int nk,
                             int64_t ld, float scale, float eps, const bf16_t* zero_page, hipStream_t s, int ds_bf16) {
  const int ntile = (nq + 15) / 16;
  const int cpi = dv_bwd_cpi(b, ntile);
  static const int xp = [] { const char* e = vitx_env("VITX_DVB_XP"); return e ? atoi(e) : 0; }();   // timing experiments (WRONG results): 1 / 2 / 4 / 8 = without stage 1 / 2 / 2b / 3
#define CALL(HT)                                                                                                                       \
  {                                                                                                                                    \
    dv_set_smem(deepvit_attn_bwd_kernel<HT>, dv_bwd_smem<HT>());                                                                       \
    hipLaunchKernelGGL(deepvit_attn_bwd_kernel<HT>, dim3(b * cpi), dim3(DVB_THREADS), dv_bwd_smem<HT>(), s, q, k, v, ldq, ldk, ldv, qb, \
                       kb, vb, d_o, ldo, ob, p_keep, w, gamma, ds_out, dq, lddq, dqb, ws, nq, nk, ld, scale, eps, zero_page, ntile, cpi, xp, ds_bf16); \
  }
  if (h == 4) CALL(4) else if (h == 8) CALL(8) else if (h == 12) CALL(12) else CALL(16)
#undef CALL
  const int nparts = b * cpi * DVB_WAVES;
  const int64_t stride = (int64_t)h * h + 2 * h;
  float* ws2 = ws + (int64_t)nparts * stride;
  launch_reduce_partials3(ws, nparts, stride, (int64_t)h * h, 1, dw, nullptr, nullptr, ws2, 1.0f, s);
  launch_reduce_partials3(ws + (int64_t)h * h, nparts, stride, h, 2, dgamma, dbeta, nullptr, ws2, 1.0f, s);
}
