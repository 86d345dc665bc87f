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
    // one thread per (query, key) point, exactly the 16 nk valid ones: the VALU work (H^2 FMAs per point) is spread evenly over
    // the waves; with 4-key groups and padded columns it took two waves on one SIMD 5 us per tile
    for (int p = tid; p < ((xp & 2) ? 0 : 16 * nk); p += DV_THREADS) {
      const int i = p / nk, j = p - i * nk;
      float y[H], vv[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) y[hh] = Pl[(hh * 16 + i) * DV_PP + j];
      // vv[g] = sum_h y[h] W[h][g], accumulated over h in ascending order for every g (the FMA chain of the backward's
      // recomputation in deepvit_point_bwd_kernel: same bits).  W is read with wave-uniform addresses (scalar loads, SGPR operands)
      // one ROW at a time, the next row requested before the current one is used; the scheduling barriers keep the compiler from
      // requesting all H rows at once -- H^2 SGPRs do not exist and it parks them in VGPR lanes (2231 v_readlane per point).
      const float* wm = w + opaque_zero();
#pragma unroll
      for (int gg = 0; gg < H; ++gg) vv[gg] = 0.f;
      float wc[H], wn[H];
#pragma unroll
      for (int gg = 0; gg < H; ++gg) wc[gg] = wm[gg];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        if (hh + 1 < H) {
#pragma unroll
          for (int gg = 0; gg < H; ++gg) wn[gg] = wm[(hh + 1) * H + gg];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gg = 0; gg < H; ++gg) vv[gg] = fmaf(y[hh], wc[gg], vv[gg]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gg = 0; gg < H; ++gg) wc[gg] = wn[gg];
      }
      float mu = 0.f;
#pragma unroll
      for (int gg = 0; gg < H; ++gg) mu += vv[gg];
      mu /= (float)H;
      float var = 0.f;
#pragma unroll
      for (int gg = 0; gg < H; ++gg) var += (vv[gg] - mu) * (vv[gg] - mu);
      const float rs = rsqrtf(var / (float)H + eps);
      const bool st = keep && (q0 + i) < nq;
      float* arow = a2_keep + (int64_t)bi * H * plane + (int64_t)(q0 + i) * ld + j;
#pragma unroll
      for (int gg = 0; gg < H; ++gg) {
        const float a2 = (vv[gg] - mu) * rs * gamma[gg] + beta[gg];
        Al[(gg * 16 + i) * DV_AP + j] = (bf16_t)a2;
        if (st) arow[gg * plane] = a2;
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

template <int H>
constexpr int dv_fwd_smem() {
  return ((H * 16 * DV_PP * 4 > 8 * DV_VBYTES) ? H * 16 * DV_PP * 4 : 8 * DV_VBYTES) + H * 16 * DV_AP * 2;
}

template <typename K>
void dv_set_smem(K kern, int bytes) {
  static const void* done[16];
  static int ndone = 0;
  for (int i = 0; i < ndone; ++i) if (done[i] == (const void*)kern) return;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (ndone < 16) done[ndone++] = (const void*)kern;
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
  if (const char* e = getenv("VITX_DV_CPI")) cpi = std::max(1, std::min(ntile, atoi(e)));   // tests: the multi-tile loop at small batches
  static const int xp = [] { const char* e = getenv("VITX_DV_XP"); return e ? atoi(e) : 0; }();   // timing experiments: 1 = no kept tensors, 2 = no stage 2, 4 = no stage 3 (WRONG results)
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
