// CaiT talking-heads attention (cait.py:109-129) as ONE forward kernel in the bf16 mode (round 5):
//   dots = q k^T * scale -> einsum('b h i j, h g -> b g i j', dots, mix_heads_pre_attn) -> softmax over keys
//   -> einsum(attn, mix_heads_post_attn) -> out = attn' v -> 'b h n d -> b n (h d)'.
// The two head mixes need every head of a (query, key) point at once, so -- as in attn_deepvit_fused.hip -- a workgroup owns 16-query tiles of one image
// for ALL heads:
//   stage 1  wave w computes S^T = K Q^T for its heads on the bf16 matrix pipe (K, Q fragments straight from the q / kv rows), scales, and leaves the
//            raw scores in LDS as fp32 [head][query][key];
//   stage 2  a wave per query row, a lane = (key slot, head quad): pre-softmax mix on the fp32 matrix pipe (mixing matrix in registers as the A operand
//            of v_mfma_f32_16x16x4_f32; see attn_headchain.hip), softmax over the keys (in-lane over the 16-key groups + a DPP row reduction),
//            post-softmax mix on the matrix pipe again, result as bf16 [head][query][key] into LDS;
//   stage 3  wave w stages V of its head into a private swizzled LDS image and computes O^T = V^T A^T with hardware-transpose reads of V.
// The [b, h, n, n] tensors touch HBM only as what the backward consumes (raw scores, softmax, mixed softmax: `keep`), written once from registers.
// Replaces three launches per block (batched QK^T GEMM, the chain kernel, batched A V GEMM) and the score tensor's round trips between them.
#include "kernels.h"
#include "attn_lds.h"

namespace {

using namespace attn_lds;

constexpr int CA_THREADS = 512;            // 8 waves
constexpr int CA_NT = 5;                   // 16-key tiles: nk <= 80
constexpr int CA_PP = 84;                  // fp32 pitch of a score row in LDS
constexpr int CA_AP = 104;                 // bf16 pitch of a mixed-softmax row (96 keys + 8)
constexpr int CA_VBYTES = 96 * ROWB;       // 12 KiB V image per wave

template <int H>
__global__ __launch_bounds__(CA_THREADS) void cait_attn_fwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t ldq, int64_t ldk, int64_t ldv,
    int64_t qb, int64_t kb, int64_t vb, bf16_t* __restrict__ o, int64_t ldo, int64_t ob, const float* __restrict__ wpre,
    const float* __restrict__ wpost, float* __restrict__ s0_keep, float* __restrict__ a1_keep, float* __restrict__ a2_keep, int keep,
    int nq, int nk, int64_t ld, float scale, const bf16_t* __restrict__ zero_page, int ntile, int cpi) {
  constexpr int HPW = (H + 7) / 8;
  constexpr int R0 = (H * 16 * CA_PP * 4 > 8 * CA_VBYTES) ? H * 16 * CA_PP * 4 : 8 * CA_VBYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* Pl = (float*)smem;                            // [H][16][CA_PP] fp32 raw scores; dead after stage 2, reused as the V images
  bf16_t* Al = (bf16_t*)(smem + R0);                   // [H][16][CA_AP] bf16
  const int xcd = blockIdx.x & 7, per = gridDim.x >> 3, rem = gridDim.x & 7;
  const int logical = xcd * per + min(xcd, rem) + (int)(blockIdx.x >> 3);
  const int bi = logical / cpi, chunk = logical - bi * cpi;
  const int t_begin = chunk * ntile / cpi, t_end = (chunk + 1) * ntile / cpi;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;
  const int nt = (nk + 15) >> 4, nu = (nk + 31) >> 5;
  const int64_t plane = (int64_t)nq * ld;

  // key columns 16 nt .. 32 nu - 1 of the mixed softmax multiply zero rows of V: zero them once (stage 2 writes columns < 16 nt only)
  for (int idx = tid; idx < H * 16 * (CA_AP / 4); idx += CA_THREADS)
    *(bf16x4*)(Al + idx * 4) = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};

  bf16x8 kf[HPW][CA_NT][2], qf[HPW][2];
#pragma unroll
  for (int s = 0; s < HPW; ++s) {
    const int head = min(wave * HPW + s, H - 1);
#pragma unroll
    for (int t = 0; t < CA_NT; ++t) {
      const int krow = min(16 * t + qi, nk - 1);
      const bf16_t* kp = k + (int64_t)bi * kb + (int64_t)krow * ldk + head * DH;
      kf[s][t][0] = *(const bf16x8*)(kp + g * 8);
      kf[s][t][1] = *(const bf16x8*)(kp + (g + 4) * 8);
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

  // stage-2 roles: key slot jl, head quad hq; the mixing matrices as A operands of the fp32 MFMA (A[m = g][k] = W[4 hq + st][g = jl])
  const int jl = lane & 15, hq = lane >> 4;
  float wa_pre[4], wa_post[4];
  bool hv[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int hh = 4 * hq + st;
    hv[st] = hh < H;
    wa_pre[st] = (hh < H && jl < H) ? wpre[hh * H + jl] : 0.f;
    wa_post[st] = (hh < H && jl < H) ? wpost[hh * H + jl] : 0.f;
  }
  auto mix4 = [](const float (&a)[4], const float (&b)[4]) {
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 4; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], b[st], d, 0, 0, 0);
    return d;
  };
#define CA_DPP(OP, v, ctrl) v = OP(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), (ctrl), 0xF, 0xF, false)))
  auto row16_sum = [](float v2) { CA_DPP(vitx_addf, v2, 0xB1); CA_DPP(vitx_addf, v2, 0x4E); CA_DPP(vitx_addf, v2, 0x124); CA_DPP(vitx_addf, v2, 0x128); return v2; };
  auto row16_max = [](float v2) { CA_DPP(fmaxf, v2, 0xB1); CA_DPP(fmaxf, v2, 0x4E); CA_DPP(fmaxf, v2, 0x124); CA_DPP(fmaxf, v2, 0x128); return v2; };
#undef CA_DPP
  // kept tensors: buffer stores through descriptors over this image's [H][nq][ld] blocks (one lane offset per row; invalid points are sent out of range)
  const int64_t img_off = (int64_t)bi * H * plane;
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(s0_keep + img_off), 0, keep ? (int)(H * plane * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a1_keep + img_off), 0, keep ? (int)(H * plane * 4) : 0, 0x00020000);
  // keep == 2: the mixed softmax is kept as bf16 [H][nq][ld2], ld2 = nk rounded up to 8 -- its only reader is the dV product, whose loader rounds it
  // to bf16 anyway (same bits, half the bytes)
  const bool a2_lp = keep == 2;
  const int ld2 = (nk + 7) & ~7;
  const int64_t plane2 = (int64_t)nq * ld2;
  const __amdgpu_buffer_rsrc_t rsA2 = a2_lp ? __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)a2_keep + (int64_t)bi * H * plane2), 0, (int)(H * plane2 * 2), 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc((void*)(a2_keep + img_off), 0, keep ? (int)(H * plane * 4) : 0, 0x00020000);
  const int plane4 = (int)plane * 4, planeA2 = a2_lp ? (int)plane2 * 2 : (int)plane * 4;

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int q0 = tile * 16;
    // ---------------------------------------------------------------- stage 1: S^T = K Q^T, scaled (cait.py:121)
#pragma unroll
    for (int s = 0; s < HPW; ++s) {
      const int head = wave * HPW + s;
      if (head < H) {
#pragma unroll
        for (int t = 0; t < CA_NT; ++t) {
          if (t < nt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = mfma16(kf[s][t][0], qf[s][0], a);       // lane: S[query qi][key 16t + 4g + r]
            a = mfma16(kf[s][t][1], qf[s][1], a);
            *(float4*)(Pl + (head * 16 + qi) * CA_PP + 16 * t + 4 * g) = make_float4(a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale);
          }
        }
      }
    }
    if (tile + 1 < t_end) load_q(tile + 1, qf);         // in flight during stages 2 and 3
    __syncthreads();

    // ---------------------------------------------------------------- stage 2: mix -> softmax -> mix (cait.py:123-125); wave w: queries w, w + 8
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const int i = wave + 8 * rr;
      const bool rv = (q0 + i) < nq;
      if (!rv) continue;   // (wave-uniform) a query row past the end: nothing of it is stored
      const int voff = (4 * hq * (int)plane + (q0 + i) * (int)ld + jl) * 4;
      float y[CA_NT][4];
      float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int gk = 0; gk < CA_NT; ++gk) {
        if (gk < nt) {
          const int j = 16 * gk + jl;
          const bool valid = j < nk;
          float x[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = Pl[(min(4 * hq + r, H - 1) * 16 + i) * CA_PP + j];
            x[r] = (valid && hv[r]) ? t : 0.f;
            // raw scores kept for the backward (dW_pre = sum S0 (x) dS1): valid points, zeros in the padding columns nk .. ld - 1
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[r]), rsS, (rv && j < ld && hv[r]) ? voff + 64 * gk : 0x7ffffff0, r * plane4, 0);
          }
          const f32x4 vv = mix4(wa_pre, x);                                 // cait.py:123
#pragma unroll
          for (int r = 0; r < 4; ++r) { y[gk][r] = vv[r]; if (valid) m[r] = fmaxf(m[r], vv[r]); }
        }
      }
      float inv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {                                         // softmax over the keys (cait.py:124)
        m[r] = row16_max(m[r]);
        float sacc = 0.f;
#pragma unroll
        for (int gk = 0; gk < CA_NT; ++gk) {
          if (gk < nt) {
            const float e = (16 * gk + jl < nk) ? __expf(y[gk][r] - m[r]) : 0.f;
            y[gk][r] = e;
            sacc += e;
          }
        }
        inv[r] = 1.0f / row16_sum(sacc);
      }
#pragma unroll
      for (int gk = 0; gk < CA_NT; ++gk) {
        if (gk < nt) {
          const int j = 16 * gk + jl;
          float p[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = y[gk][r] * inv[r];
          const f32x4 z = mix4(wa_post, p);                                 // cait.py:125
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int off = (rv && j < ld && hv[r]) ? voff + 64 * gk : 0x7ffffff0;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p[r]), rsA1, off, r * plane4, 0);
            if (a2_lp)
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)(j < nk ? z[r] : 0.f)), rsA2,
                                                    (j < ld2 && hv[r]) ? (4 * hq * (int)plane2 + (q0 + i) * ld2 + j) * 2 : 0x7ffffff0, r * planeA2, 0);
            else
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, j < nk ? z[r] : 0.f), rsA2, off, r * planeA2, 0);
            if (hv[r]) Al[((4 * hq + r) * 16 + i) * CA_AP + j] = (j < nk) ? (bf16_t)z[r] : (bf16_t)0.f;   // (keys nk .. 16 nt - 1 multiply zero rows of V)
          }
        }
      }
    }
    __syncthreads();

    // ---------------------------------------------------------------- stage 3: out = attn' v, merged heads (cait.py:127-128)
    if (o != nullptr) {   // (o == nullptr: the backward of a block whose score tensors were not kept recomputes them with this kernel: stages 1-2 only)
      char* vbuf = smem + wave * CA_VBYTES;
#pragma unroll
      for (int s = 0; s < HPW; ++s) {
        const int head = wave * HPW + s;
        if (head >= H) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous head's transpose reads have returned
        stage_head_dma(v + (int64_t)bi * vb + head * DH, ldv, nk, 32 * nu, vbuf, zero_page, 0, lane, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 oacc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int u = 0; u < nu; ++u) {
          const bf16_t* ar = Al + (head * 16 + qi) * CA_AP + 32 * u + 4 * g;
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
    __syncthreads();   // the V images overlay the score buffer of the next tile
  }
}

template <int H>
constexpr int ca_fwd_smem() {
  return ((H * 16 * CA_PP * 4 > 8 * CA_VBYTES) ? H * 16 * CA_PP * 4 : 8 * CA_VBYTES) + H * 16 * CA_AP * 2;
}

}  // namespace

bool cait_attn_fused_supported(int h, int dim_head, int nq, int nk) {
  return dim_head == DH && (h == 4 || h == 8 || h == 12 || h == 16) && nq >= 16 && nk >= 1 && nk <= 16 * CA_NT;
}

void launch_cait_attn_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv, int64_t qb, int64_t kb, int64_t vb,
                          bf16_t* o, int64_t ldo, int64_t ob, const float* wpre, const float* wpost, float* s0_keep, float* a1_keep, float* a2_keep,
                          int keep, int b, int h, int nq, int nk, int64_t ld, float scale, const bf16_t* zero_page, hipStream_t s) {
  const int ntile = (nq + 15) / 16;
  const int cpi = b >= 192 ? 1 : std::max(1, std::min(ntile, (256 + b - 1) / b));
#define CALL(HT)                                                                                                                      \
  {                                                                                                                                   \
    vitx_set_max_smem((const void*)cait_attn_fwd_kernel<HT>, ca_fwd_smem<HT>());                                                      \
    hipLaunchKernelGGL(cait_attn_fwd_kernel<HT>, dim3(b * cpi), dim3(CA_THREADS), ca_fwd_smem<HT>(), s, q, k, v, ldq, ldk, ldv, qb, kb, vb, o, ldo,   \
                       ob, wpre, wpost, s0_keep, a1_keep, a2_keep, keep, nq, nk, ld, scale, zero_page, ntile, cpi);                    \
  }
  if (h == 4) CALL(4) else if (h == 8) CALL(8) else if (h == 12) CALL(12) else CALL(16)
#undef CALL
}
