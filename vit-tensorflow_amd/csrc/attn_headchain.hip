// Fused head-axis chains of the materialised attention variants (one pass over the score tensor instead of three):
//   CaiT talking heads      (cait.py:121-125):  S0 -> mix(W_pre) -> softmax_j -> A1 -> mix(W_post) -> A2
//   DeepViT Re-attention    (deepvit.py:79-84): S0 -> softmax_j -> A0 -> mix(W_re) -> M -> LayerNorm over heads -> A2
// and their VJPs (mixing-matrix, LayerNorm gamma/beta gradients included).  Score tensors are [b, h, nq, ld] fp32, ld = round_up(nk,4).
// Mapping: ONE WAVE PER (image, query) ROW.  Lanes run over the keys j (two passes cover nk <= 128), every lane keeps the values of
// all h heads of its key in registers, so
//   * the h x h mixes are in-lane FMAs against an LDS-broadcast weight matrix,
//   * the softmax over j is h wave reductions,
//   * the LayerNorm over heads is in-lane,
//   * the mixing-matrix gradient dW[h][g] = sum over (row, j) of in[h] * dout[g] is a [16 x points] x [points x 16] product that goes
//     to the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, exact fp32 products): the per-lane vectors are transposed into MFMA operand
//     layout through a per-wave LDS scratch.  Per-wave partial sums are written out and reduced in fixed order (deterministic).
// Arithmetic order inside each op is the same as in the unfused kernels of attn_generic.hip (same expf, same FMA chains), so the
// fp32 parity mode is unaffected.  Head counts 4/8/12/16 (padded to 16 for the MFMA), nk <= 64; other shapes use the unfused kernels.
#include "kernels.h"

namespace {

constexpr int HC_WAVES = 4;          // waves (rows in flight) per workgroup
constexpr int HC_PITCH = 20;         // floats per point in the transpose scratch (16 + pad, 16-B aligned rows)
constexpr int HC_BLOCKS = 1024;      // upper bound on workgroups (partials: HC_BLOCKS * HC_WAVES rows)

template <int H>
__device__ __forceinline__ void load_w(float* dst, const float* __restrict__ w) {
  for (int i = threadIdx.x; i < H * H; i += blockDim.x) dst[i] = w[i];
}

// dW[hh][g] += sum over this wave's 64 points of xv[hh] * dv[g]   (acc = MFMA D fragment: hh = 4*(lane>>4)+r, g = lane&15)
template <int H>
__device__ __forceinline__ void outer_accumulate(f32x4& acc, const float (&xv)[H], const float (&dv)[H], float* xs, float* ys, int lane) {
  float* xr = xs + lane * HC_PITCH;
  float* yr = ys + lane * HC_PITCH;
#pragma unroll
  for (int k = 0; k < 16; k += 4) {
    *(float4*)(xr + k) = make_float4(k + 0 < H ? xv[k + 0 < H ? k + 0 : 0] : 0.f, k + 1 < H ? xv[k + 1 < H ? k + 1 : 0] : 0.f,
                                     k + 2 < H ? xv[k + 2 < H ? k + 2 : 0] : 0.f, k + 3 < H ? xv[k + 3 < H ? k + 3 : 0] : 0.f);
    *(float4*)(yr + k) = make_float4(k + 0 < H ? dv[k + 0 < H ? k + 0 : 0] : 0.f, k + 1 < H ? dv[k + 1 < H ? k + 1 : 0] : 0.f,
                                     k + 2 < H ? dv[k + 2 < H ? k + 2 : 0] : 0.f, k + 3 < H ? dv[k + 3 < H ? k + 3 : 0] : 0.f);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS traffic is processed in order: writes are visible to the reads below
  const int m = lane & 15, kp = lane >> 4;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float a = xs[(4 * s + kp) * HC_PITCH + m];
    const float bq = ys[(4 * s + kp) * HC_PITCH + m];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc, 0, 0, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int H>
__device__ __forceinline__ void store_dw(const f32x4& acc, float* dst, int lane) {   // dst [H][H]
  const int g = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int hh = 4 * (lane >> 4) + r;
    if (hh < H && g < H) dst[hh * H + g] = acc[r];
  }
}

// ------------------------------------------------------------------------------------------------ CaiT
template <int H, int NP>
__global__ __launch_bounds__(64 * HC_WAVES) void cait_chain_fwd_kernel(const float* __restrict__ s0, const float* __restrict__ wpre,
                                                                       const float* __restrict__ wpost, float* __restrict__ a1,
                                                                       float* __restrict__ a2, int64_t rows, int nq, int nk, int64_t ld) {
  // (mixing matrices are read from global memory with wave-uniform addresses -> scalar loads / SGPR operands; staged through LDS
  //  the compiler hoists them into 2 x 256 VGPRs per lane and the kernel drops to one or two waves per SIMD)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t plane = (int64_t)nq * ld;
  for (int64_t row = (int64_t)blockIdx.x * HC_WAVES + wave; row < rows; row += (int64_t)gridDim.x * HC_WAVES) {
    const int64_t bi = row / nq, i = row - bi * nq;
    const int64_t base = bi * H * plane + i * ld;
    const float* wpre_r = wpre + opaque_zero();
    const float* wpost_r = wpost + opaque_zero();
    float y[NP][H];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int j = p * 64 + lane;
      const bool valid = j < nk;
      float x[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) x[hh] = valid ? s0[base + (int64_t)hh * plane + j] : 0.f;
#pragma unroll
      for (int g = 0; g < H; ++g) y[p][g] = 0.f;
#pragma unroll
      for (int hh = 0; hh < H; ++hh)     // one contiguous row of W (16 scalars) live at a time; every output still sums hh ascending
#pragma unroll
        for (int g = 0; g < H; ++g) y[p][g] = fmaf(x[hh], wpre_r[hh * H + g], y[p][g]);
    }
#pragma unroll
    for (int g = 0; g < H; ++g) {   // softmax over the keys (cait.py:124), same operation order as softmax_rows_kernel
      float m = -INFINITY;
#pragma unroll
      for (int p = 0; p < NP; ++p) if (p * 64 + lane < nk) m = fmaxf(m, y[p][g]);
      m = wave_max(m);
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) { const float e = (p * 64 + lane < nk) ? expf(y[p][g] - m) : 0.f; y[p][g] = e; s += e; }
      s = wave_sum(s);
      const float inv = 1.0f / s;
#pragma unroll
      for (int p = 0; p < NP; ++p) y[p][g] *= inv;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int j = p * 64 + lane;
      if (j >= nk) continue;
      if (a1 != nullptr) {
#pragma unroll
        for (int g = 0; g < H; ++g) a1[base + (int64_t)g * plane + j] = y[p][g];
      }
      float z[H];
#pragma unroll
      for (int g = 0; g < H; ++g) z[g] = 0.f;
#pragma unroll
      for (int hh = 0; hh < H; ++hh)
#pragma unroll
        for (int g = 0; g < H; ++g) z[g] = fmaf(y[p][hh], wpost_r[hh * H + g], z[g]);
#pragma unroll
      for (int g = 0; g < H; ++g) a2[base + (int64_t)g * plane + j] = z[g];
    }
  }
}

// d(A2) in `da` -> d(S0) written back into `da`; partial[wave][0] = dW_post, partial[wave][1] = dW_pre
template <int H, int NP>
__global__ __launch_bounds__(64 * HC_WAVES) void cait_chain_bwd_kernel(const float* __restrict__ s0, const float* __restrict__ a1,
                                                                       float* __restrict__ da, const float* __restrict__ wpre,
                                                                       const float* __restrict__ wpost, float* __restrict__ partial,
                                                                       int64_t rows, int nq, int nk, int64_t ld) {
  __shared__ float ws[2][H * H];
  __shared__ __attribute__((aligned(16))) float scratch[HC_WAVES][2][64 * HC_PITCH];
  load_w<H>(ws[0], wpre);
  load_w<H>(ws[1], wpost);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xs = scratch[wave][0];
  float* ys = scratch[wave][1];
  const int64_t plane = (int64_t)nq * ld;
  f32x4 acc_post = {0.f, 0.f, 0.f, 0.f}, acc_pre = {0.f, 0.f, 0.f, 0.f};
  for (int64_t row = (int64_t)blockIdx.x * HC_WAVES + wave; row < rows; row += (int64_t)gridDim.x * HC_WAVES) {
    const int64_t bi = row / nq, i = row - bi * nq;
    const int64_t base = bi * H * plane + i * ld;
    const float* wpre_r = wpre + opaque_zero();
    const float* wpost_r = wpost + opaque_zero();
    float av[NP][H], d1[NP][H];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int j = p * 64 + lane;
      const bool valid = j < nk;
      float dz[H];
#pragma unroll
      for (int g = 0; g < H; ++g) {
        av[p][g] = valid ? a1[base + (int64_t)g * plane + j] : 0.f;
        dz[g] = valid ? da[base + (int64_t)g * plane + j] : 0.f;
      }
      outer_accumulate<H>(acc_post, av[p], dz, xs, ys, lane);          // dW_post[hh][g] += A1[hh] * dA2[g]
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < H; ++g) a = fmaf(dz[g], wpost_r[hh * H + g], a);
        d1[p][hh] = a;                                                 // dA1
      }
    }
#pragma unroll
    for (int g = 0; g < H; ++g) {                                      // softmax VJP: dS1 = A1 * (dA1 - sum_j A1 dA1)
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) s += av[p][g] * d1[p][g];
      s = wave_sum(s);
#pragma unroll
      for (int p = 0; p < NP; ++p) d1[p][g] = av[p][g] * (d1[p][g] - s);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int j = p * 64 + lane;
      const bool valid = j < nk;
      float x[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) x[hh] = valid ? s0[base + (int64_t)hh * plane + j] : 0.f;
      outer_accumulate<H>(acc_pre, x, d1[p], xs, ys, lane);            // dW_pre[hh][g] += S0[hh] * dS1[g]
      if (valid) {
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
          float a = 0.f;
#pragma unroll
          for (int g = 0; g < H; ++g) a = fmaf(d1[p][g], wpre_r[hh * H + g], a);
          da[base + (int64_t)hh * plane + j] = a;                      // dS0
        }
      }
    }
  }
  float* pw = partial + ((int64_t)blockIdx.x * HC_WAVES + wave) * 2 * H * H;
  store_dw<H>(acc_post, pw, lane);
  store_dw<H>(acc_pre, pw + H * H, lane);
}


// ------------------------------------------------------------------------------------------------ CaiT, head mixes on the fp32 matrix pipe (round 5)
// Same chains, other lane roles: a wave still owns one (image, query) row, but a lane is (key slot jl = lane & 15, head quad hq = lane >> 4) and holds
// the values of heads 4 hq + r (r < 4) of its key, a row being NG groups of 16 keys.  Both h x h mixes of a group are then four
// v_mfma_f32_16x16x4_f32 each (D[m][n] += sum_k A[m][k] B[k][n], n = the group's 16 keys, k <-> the head a lane holds in register st) with the mixing
// matrix sitting in registers as the A operand:
//   out[key][g] = sum_h in[key][h] W[h][g]   (forward mixes):  A[m = g][k] = W[4 hq + st][g = jl],  B = in  of head 4 hq + st
//   din[key][h] = sum_g dout[key][g] W[h][g] (their VJPs):     A[m = h][k] = W[h = jl][4 hq + st],  B = dout of head 4 hq + st
// and the result lands in the same ownership (D: lane (n = jl, rows 4 hq + r)).  The fp32 MFMA multiplies exactly and accumulates in fp32 like the
// FMA chains it replaces (other summation order: 1e-7 relative).  Why: as in-lane FMAs the mixes need all H values of a key in one lane and the
// matrix re-read through scalar loads for every row (2 x H row round trips of ~200 cycles, H-element register arrays); here a lane holds 4 values
// per tensor, the matrices cost 8-16 registers, and the kernels run at the speed of their three / four [b, h, n, n] fp32 streams.
// The softmax over the keys = in-lane over the groups + a 16-lane DPP row reduction; everything the VJP needs stays in registers.
template <int H>
__device__ __forceinline__ void mix_operands(const float* __restrict__ w, int jl, int hq, float (&wa)[4], float (&wb)[4]) {
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int hh = 4 * hq + st;
    wa[st] = (hh < H && jl < H) ? w[hh * H + jl] : 0.f;
    wb[st] = (hh < H && jl < H) ? w[jl * H + hh] : 0.f;
  }
}
__device__ __forceinline__ f32x4 mix4(const float (&a)[4], const float (&b)[4]) {
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < 4; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], b[st], d, 0, 0, 0);
  return d;
}
#define HC_DPP(OP, v, ctrl) v = OP(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), (ctrl), 0xF, 0xF, false)))
__device__ __forceinline__ float row16_sum(float v) {   // over the 16 key slots of a head quad (one DPP row); every lane of the row gets the total
  HC_DPP(vitx_addf, v, 0xB1); HC_DPP(vitx_addf, v, 0x4E); HC_DPP(vitx_addf, v, 0x124); HC_DPP(vitx_addf, v, 0x128);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  HC_DPP(fmaxf, v, 0xB1); HC_DPP(fmaxf, v, 0x4E); HC_DPP(fmaxf, v, 0x124); HC_DPP(fmaxf, v, 0x128);
  return v;
}
#undef HC_DPP
// dW[hh][g] += sum over a group's 16 keys of xv[hh] dv[g] (xv / dv: this lane's 4 heads of its key), transposed through the per-wave scratch
__device__ __forceinline__ void outer_accumulate16(f32x4& acc, const float (&xv)[4], const float (&dv)[4], float* xs, float* ys, int jl, int hq) {
  *(float4*)(xs + jl * HC_PITCH + 4 * hq) = make_float4(xv[0], xv[1], xv[2], xv[3]);
  *(float4*)(ys + jl * HC_PITCH + 4 * hq) = make_float4(dv[0], dv[1], dv[2], dv[3]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float ta[4], tb[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) { ta[st] = xs[(4 * st + hq) * HC_PITCH + jl]; tb[st] = ys[(4 * st + hq) * HC_PITCH + jl]; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[st], tb[st], acc, 0, 0, 0);
}

constexpr int HC_NG = 4;   // 16-key groups of a row: nk <= 64.  Group c = the keys 4 jl + c, so that a lane's four groups are ONE 16-byte access per head and
                           // tensor (16 lanes x 16 B = a contiguous 256-B row piece per instruction instead of four 64-B pieces)

template <int H>
__global__ __launch_bounds__(64 * HC_WAVES) void cait_chain_fwd_mfma_kernel(const float* __restrict__ s0, const float* __restrict__ wpre,
                                                                            const float* __restrict__ wpost, float* __restrict__ a1,
                                                                            float* __restrict__ a2, int64_t rows, int nq, int nk, int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, jl = lane & 15, hq = lane >> 4;
  const int64_t plane = (int64_t)nq * ld;
  const bool in_row = 4 * jl < ld;                     // this lane's 16 bytes lie inside the row
  const int jo = in_row ? 4 * jl : 0;                  // (loads are unconditional from a clamped address and masked afterwards)
  float wa_pre[4], wa_post[4], unused[4];
  mix_operands<H>(wpre, jl, hq, wa_pre, unused);
  mix_operands<H>(wpost, jl, hq, wa_post, unused);
  bool hv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) hv[r] = 4 * hq + r < H;
  for (int64_t row = (int64_t)blockIdx.x * HC_WAVES + wave; row < rows; row += (int64_t)gridDim.x * HC_WAVES) {
    const int64_t bi = row / nq, i = row - bi * nq;
    const int64_t base0 = bi * H * plane + i * ld + jo;
    float4 xin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xin[r] = *(const float4*)(s0 + base0 + (int64_t)min(4 * hq + r, H - 1) * plane);
    float y[HC_NG][4];
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int c = 0; c < HC_NG; ++c) {
      const bool valid = 4 * jl + c < nk;
      float x[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float t = c == 0 ? xin[r].x : (c == 1 ? xin[r].y : (c == 2 ? xin[r].z : xin[r].w)); x[r] = (valid && hv[r]) ? t : 0.f; }
      const f32x4 v = mix4(wa_pre, x);                                   // cait.py:123
#pragma unroll
      for (int r = 0; r < 4; ++r) { y[c][r] = v[r]; if (valid) m[r] = fmaxf(m[r], v[r]); }
    }
    float inv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                        // softmax over the keys (cait.py:124)
      m[r] = row16_max(m[r]);
      float sacc = 0.f;
#pragma unroll
      for (int c = 0; c < HC_NG; ++c) {
        const float e = (4 * jl + c < nk) ? expf(y[c][r] - m[r]) : 0.f;
        y[c][r] = e;
        sacc += e;
      }
      inv[r] = 1.0f / row16_sum(sacc);
    }
    float p[HC_NG][4], z[HC_NG][4];
#pragma unroll
    for (int c = 0; c < HC_NG; ++c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) p[c][r] = y[c][r] * inv[r];
      const f32x4 v = mix4(wa_post, p[c]);                               // cait.py:125
#pragma unroll
      for (int r = 0; r < 4; ++r) z[c][r] = v[r];
    }
    if (in_row) {                                                        // (columns nk .. ld - 1: zeros)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (hv[r]) {
          const int64_t o = base0 + (int64_t)(4 * hq + r) * plane;
          if (a1 != nullptr) *(float4*)(a1 + o) = make_float4(p[0][r], p[1][r], p[2][r], p[3][r]);
          *(float4*)(a2 + o) = make_float4(z[0][r], z[1][r], z[2][r], z[3][r]);
        }
      }
    }
  }
}

// d(A2) in `da` -> d(S0) written back into `da`; partial[wave][0] = dW_post, partial[wave][1] = dW_pre
template <int H>
__global__ __launch_bounds__(64 * HC_WAVES) void cait_chain_bwd_mfma_kernel(const float* __restrict__ s0, const float* __restrict__ a1,
                                                                            float* __restrict__ da, const float* __restrict__ wpre,
                                                                            const float* __restrict__ wpost, float* __restrict__ partial,
                                                                            int64_t rows, int nq, int nk, int64_t ld, bf16_t* __restrict__ ds_lp) {
  __shared__ __attribute__((aligned(16))) float scratch[HC_WAVES][2][16 * HC_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, jl = lane & 15, hq = lane >> 4;
  float* xs = scratch[wave][0];
  float* ys = scratch[wave][1];
  const int64_t plane = (int64_t)nq * ld;
  // ds_lp != null (round 5): d(dots) goes out as bf16 [H][nq][ld2] planes (ld2 = nk rounded up to 8) into a buffer of its own instead of over d(attn') in
  // fp32 -- its only readers are the dQ / dK products, whose loaders round it to bf16 anyway
  const int ld2 = (nk + 7) & ~7;
  const int64_t plane2 = (int64_t)nq * ld2;
  const bool in_row = 4 * jl < ld;
  const int jo = in_row ? 4 * jl : 0;
  float wb_pre[4], wb_post[4], unused[4];
  mix_operands<H>(wpre, jl, hq, unused, wb_pre);
  mix_operands<H>(wpost, jl, hq, unused, wb_post);
  bool hv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) hv[r] = 4 * hq + r < H;
  f32x4 acc_post = {0.f, 0.f, 0.f, 0.f}, acc_pre = {0.f, 0.f, 0.f, 0.f};
  auto comp = [](const float4& t, int c) { return c == 0 ? t.x : (c == 1 ? t.y : (c == 2 ? t.z : t.w)); };
  for (int64_t row = (int64_t)blockIdx.x * HC_WAVES + wave; row < rows; row += (int64_t)gridDim.x * HC_WAVES) {
    const int64_t bi = row / nq, i = row - bi * nq;
    const int64_t base0 = bi * H * plane + i * ld + jo;
    float4 a1in[4], dain[4], s0in[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t o = base0 + (int64_t)min(4 * hq + r, H - 1) * plane;
      a1in[r] = *(const float4*)(a1 + o);
      dain[r] = *(const float4*)(da + o);
      s0in[r] = *(const float4*)(s0 + o);
    }
    float av[HC_NG][4], d1[HC_NG][4];
    float sdot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < HC_NG; ++c) {
      const bool valid = 4 * jl + c < nk;
      float dz[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        av[c][r] = (valid && hv[r]) ? comp(a1in[r], c) : 0.f;
        dz[r] = (valid && hv[r]) ? comp(dain[r], c) : 0.f;
      }
      outer_accumulate16(acc_post, av[c], dz, xs, ys, jl, hq);           // dW_post[hh][g] += A1[hh] * dA2[g]
      const f32x4 v = mix4(wb_post, dz);                                  // dA1
#pragma unroll
      for (int r = 0; r < 4; ++r) { d1[c][r] = v[r]; sdot[r] += av[c][r] * v[r]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sdot[r] = row16_sum(sdot[r]);             // softmax VJP: dS1 = A1 * (dA1 - sum_j A1 dA1)
    float dso[HC_NG][4];
#pragma unroll
    for (int c = 0; c < HC_NG; ++c) {
      const bool valid = 4 * jl + c < nk;
      float ds1[4], x0[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { ds1[r] = av[c][r] * (d1[c][r] - sdot[r]); x0[r] = (valid && hv[r]) ? comp(s0in[r], c) : 0.f; }
      outer_accumulate16(acc_pre, x0, ds1, xs, ys, jl, hq);              // dW_pre[hh][g] += S0[hh] * dS1[g]
      const f32x4 v = mix4(wb_pre, ds1);                                 // dS0
#pragma unroll
      for (int r = 0; r < 4; ++r) dso[c][r] = valid ? v[r] : 0.f;
    }
    if (ds_lp) {
      if (4 * jl < ld2) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (hv[r]) *(bf16x4*)(ds_lp + (bi * H + 4 * hq + r) * plane2 + i * ld2 + 4 * jl) = bf16x4{(bf16_t)dso[0][r], (bf16_t)dso[1][r], (bf16_t)dso[2][r], (bf16_t)dso[3][r]};
      }
    } else if (in_row) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (hv[r]) *(float4*)(da + base0 + (int64_t)(4 * hq + r) * plane) = make_float4(dso[0][r], dso[1][r], dso[2][r], dso[3][r]);
    }
  }
  float* pw = partial + ((int64_t)blockIdx.x * HC_WAVES + wave) * 2 * H * H;
  store_dw<H>(acc_post, pw, lane);
  store_dw<H>(acc_pre, pw + H * H, lane);
}

// ------------------------------------------------------------------------------------------------ DeepViT
template <int H, int NP>
__global__ __launch_bounds__(64 * HC_WAVES) void deepvit_chain_fwd_kernel(float* __restrict__ s0 /* in: scores; out: softmax (if keep) */,
                                                                          const float* __restrict__ wre, const float* __restrict__ gamma,
                                                                          const float* __restrict__ beta, float* __restrict__ mixed /* nullable */,
                                                                          float* __restrict__ a2, int keep, int64_t rows, int nq, int nk,
                                                                          int64_t ld, float eps) {
  // (mixing matrix via scalar loads, see the CaiT kernels)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t plane = (int64_t)nq * ld;
  float gm[H], bt[H];
#pragma unroll
  for (int g = 0; g < H; ++g) { gm[g] = gamma[g]; bt[g] = beta[g]; }
  for (int64_t row = (int64_t)blockIdx.x * HC_WAVES + wave; row < rows; row += (int64_t)gridDim.x * HC_WAVES) {
    const int64_t bi = row / nq, i = row - bi * nq;
    const int64_t base = bi * H * plane + i * ld;
    const float* wre_r = wre + opaque_zero();
    float y[NP][H];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int hh = 0; hh < H; ++hh) y[p][hh] = (p * 64 + lane < nk) ? s0[base + (int64_t)hh * plane + p * 64 + lane] : 0.f;
#pragma unroll
    for (int g = 0; g < H; ++g) {   // softmax over the keys (deepvit.py:80)
      float m = -INFINITY;
#pragma unroll
      for (int p = 0; p < NP; ++p) if (p * 64 + lane < nk) m = fmaxf(m, y[p][g]);
      m = wave_max(m);
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) { const float e = (p * 64 + lane < nk) ? expf(y[p][g] - m) : 0.f; y[p][g] = e; s += e; }
      s = wave_sum(s);
      const float inv = 1.0f / s;
#pragma unroll
      for (int p = 0; p < NP; ++p) y[p][g] *= inv;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int j = p * 64 + lane;
      if (j >= nk) continue;
      float v[H];
      float mu = 0.f;
#pragma unroll
      for (int g = 0; g < H; ++g) {
        if (keep) s0[base + (int64_t)g * plane + j] = y[p][g];
        float a = 0.f;
#pragma unroll
        for (int hh = 0; hh < H; ++hh) a = fmaf(y[p][hh], wre_r[hh * H + g], a);   // re-attention mix (deepvit.py:83)
        v[g] = a;
      }
#pragma unroll
      for (int g = 0; g < H; ++g) { if (mixed != nullptr) mixed[base + (int64_t)g * plane + j] = v[g]; mu += v[g]; }
      mu /= (float)H;
      float var = 0.f;
#pragma unroll
      for (int g = 0; g < H; ++g) var += (v[g] - mu) * (v[g] - mu);
      const float rs = rsqrtf(var / (float)H + eps);
#pragma unroll
      for (int g = 0; g < H; ++g) a2[base + (int64_t)g * plane + j] = (v[g] - mu) * rs * gm[g] + bt[g];   // LayerNorm over heads (deepvit.py:84)
    }
  }
}

// d(A2) in `da` -> d(S0) written back; partial[wave] = [dW_re (H*H) | dgamma (H) | dbeta (H)]
template <int H, int NP>
__global__ __launch_bounds__(64 * HC_WAVES) void deepvit_chain_bwd_kernel(const float* __restrict__ a0, const float* __restrict__ mixed,
                                                                          float* __restrict__ da, const float* __restrict__ wre,
                                                                          const float* __restrict__ gamma, float* __restrict__ partial,
                                                                          int64_t rows, int nq, int nk, int64_t ld, float eps) {
  __shared__ float ws[H * H];
  __shared__ __attribute__((aligned(16))) float scratch[HC_WAVES][2][64 * HC_PITCH];
  load_w<H>(ws, wre);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xs = scratch[wave][0];
  float* ys = scratch[wave][1];
  const int64_t plane = (int64_t)nq * ld;
  float gm[H], ag[H], ab[H];
#pragma unroll
  for (int g = 0; g < H; ++g) { gm[g] = gamma[g]; ag[g] = 0.f; ab[g] = 0.f; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t row = (int64_t)blockIdx.x * HC_WAVES + wave; row < rows; row += (int64_t)gridDim.x * HC_WAVES) {
    const int64_t bi = row / nq, i = row - bi * nq;
    const int64_t base = bi * H * plane + i * ld;
    const float* wre_r = wre + opaque_zero();
    float av[NP][H], d0[NP][H];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int j = p * 64 + lane;
      const bool valid = j < nk;
      float xh[H], dm[H];
      float mu = 0.f;
#pragma unroll
      for (int g = 0; g < H; ++g) {
        av[p][g] = valid ? a0[base + (int64_t)g * plane + j] : 0.f;
        xh[g] = valid ? mixed[base + (int64_t)g * plane + j] : 0.f;
        mu += xh[g];
      }
      mu /= (float)H;
      float var = 0.f;
#pragma unroll
      for (int g = 0; g < H; ++g) { xh[g] -= mu; var += xh[g] * xh[g]; }
      const float rs = rsqrtf(var / (float)H + eps);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int g = 0; g < H; ++g) {                                    // LayerNorm-over-heads VJP (same order as headnorm_bwd_kernel)
        xh[g] *= rs;
        const float d = valid ? da[base + (int64_t)g * plane + j] : 0.f;
        ab[g] += d;
        ag[g] += d * xh[g];
        dm[g] = d * gm[g];
        s1 += dm[g];
        s2 += dm[g] * xh[g];
      }
      s1 /= (float)H;
      s2 /= (float)H;
#pragma unroll
      for (int g = 0; g < H; ++g) dm[g] = valid ? rs * (dm[g] - s1 - xh[g] * s2) : 0.f;
      outer_accumulate<H>(acc, av[p], dm, xs, ys, lane);               // dW_re[hh][g] += A0[hh] * dM[g]
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < H; ++g) a = fmaf(dm[g], wre_r[hh * H + g], a);
        d0[p][hh] = a;                                                 // dA0
      }
    }
#pragma unroll
    for (int g = 0; g < H; ++g) {                                      // softmax VJP
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) s += av[p][g] * d0[p][g];
      s = wave_sum(s);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int j = p * 64 + lane;
        if (j < nk) da[base + (int64_t)g * plane + j] = av[p][g] * (d0[p][g] - s);
      }
    }
  }
  float* pw = partial + ((int64_t)blockIdx.x * HC_WAVES + wave) * (H * H + 2 * H);
  store_dw<H>(acc, pw, lane);
#pragma unroll
  for (int g = 0; g < H; ++g) {
    const float sg = wave_sum(ag[g]), sb = wave_sum(ab[g]);
    if (lane == 0) { pw[H * H + g] = sg; pw[H * H + H + g] = sb; }
  }
}

int chain_blocks(int64_t rows) { return (int)std::max<int64_t>(1, std::min<int64_t>(HC_BLOCKS, ceil_div(rows, HC_WAVES))); }

}  // namespace

// nk <= 64 (one key per lane, everything in registers).  For longer rows a row-per-wave kernel has to hold several passes of every
// per-head vector (spills: 16.6 ms per DeepViT step at 65 keys) or walk the row in several sweeps (12.4 ms) -- both lose to the unfused
// point-per-thread kernels (7.8 ms), which have far more loads in flight; those rows keep the unfused kernels.
bool headchain_supported(int h, int nk) { return (h == 4 || h == 8 || h == 12 || h == 16) && nk >= 1 && nk <= 64; }
int64_t headchain_ws_elems(int h) { return (int64_t)(HC_BLOCKS * HC_WAVES + 40) * (2 * h * h + 2 * h); }

#define HC_DISPATCH(h, CALL) \
  do { if (h == 16) { CALL(16); } else if (h == 12) { CALL(12); } else if (h == 8) { CALL(8); } else { CALL(4); } } while (0)

void launch_cait_chain_fwd(const float* s0, const float* wpre, const float* wpost, float* a1_or_null, float* a2, int b, int h, int nq, int nk,
                           int64_t ld, hipStream_t s) {
  const int64_t rows = (int64_t)b * nq;
  const dim3 grid(chain_blocks(rows)), block(64 * HC_WAVES);
  static const int mfma_mix = [] { const char* v = vitx_env("VITX_CHAIN_MFMA"); return v ? atoi(v) : 1; }();   // 0: the in-lane FMA form (A/B reference)
  if (mfma_mix) {
#define CALL(H) hipLaunchKernelGGL((cait_chain_fwd_mfma_kernel<H>), grid, block, 0, s, s0, wpre, wpost, a1_or_null, a2, rows, nq, nk, ld)
    HC_DISPATCH(h, CALL);
#undef CALL
    return;
  }
#define CALL(H) hipLaunchKernelGGL((cait_chain_fwd_kernel<H, 1>), grid, block, 0, s, s0, wpre, wpost, a1_or_null, a2, rows, nq, nk, ld)
  HC_DISPATCH(h, CALL);
#undef CALL
}
static int chain_mfma_mix() {
  static const int v = [] { const char* e = vitx_env("VITX_CHAIN_MFMA"); return e ? atoi(e) : 1; }();
  return v;
}
bool cait_chain_bwd_bf16_out_ok() { return chain_mfma_mix() != 0; }   // (only the MFMA form of the kernel has the bf16 output)
void launch_cait_chain_bwd(const float* s0, const float* a1, float* da_inout, const float* wpre, const float* wpost, float* ws, float* dwpre,
                           float* dwpost, int b, int h, int nq, int nk, int64_t ld, hipStream_t s, bf16_t* ds_lp) {
  const int64_t rows = (int64_t)b * nq;
  const int nblk = chain_blocks(rows);
  const dim3 grid(nblk), block(64 * HC_WAVES);
  const int mfma_mix = chain_mfma_mix();
  if (mfma_mix) {
#define CALL(H) hipLaunchKernelGGL((cait_chain_bwd_mfma_kernel<H>), grid, block, 0, s, s0, a1, da_inout, wpre, wpost, ws, rows, nq, nk, ld, ds_lp)
    HC_DISPATCH(h, CALL);
#undef CALL
  } else {
#define CALL(H) hipLaunchKernelGGL((cait_chain_bwd_kernel<H, 1>), grid, block, 0, s, s0, a1, da_inout, wpre, wpost, ws, rows, nq, nk, ld)
    HC_DISPATCH(h, CALL);
#undef CALL
  }
  const int nparts = nblk * HC_WAVES;
  const int64_t stride = 2 * (int64_t)h * h;
  float* ws2 = ws + (int64_t)nparts * stride;
  launch_reduce_partials3(ws, nparts, stride, (int64_t)h * h, 2, dwpost, dwpre, nullptr, ws2, 1.0f, s);
}
void launch_deepvit_chain_fwd(float* s0_inout, const float* wre, const float* gamma, const float* beta, float* mixed_or_null, float* a2, int keep,
                              int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s) {
  const int64_t rows = (int64_t)b * nq;
  const dim3 grid(chain_blocks(rows)), block(64 * HC_WAVES);
#define CALL(H) hipLaunchKernelGGL((deepvit_chain_fwd_kernel<H, 1>), grid, block, 0, s, s0_inout, wre, gamma, beta, mixed_or_null, a2, keep, rows, nq, nk, ld, eps)
  HC_DISPATCH(h, CALL);
#undef CALL
}
void launch_deepvit_chain_bwd(const float* a0, const float* mixed, float* da_inout, const float* wre, const float* gamma, float* ws, float* dwre,
                              float* dgamma, float* dbeta, int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s) {
  const int64_t rows = (int64_t)b * nq;
  const int nblk = chain_blocks(rows);
  const dim3 grid(nblk), block(64 * HC_WAVES);
#define CALL(H) hipLaunchKernelGGL((deepvit_chain_bwd_kernel<H, 1>), grid, block, 0, s, a0, mixed, da_inout, wre, gamma, ws, rows, nq, nk, ld, eps)
  HC_DISPATCH(h, CALL);
#undef CALL
  const int nparts = nblk * HC_WAVES;
  const int64_t stride = (int64_t)h * h + 2 * h;
  float* ws2 = ws + (int64_t)nparts * stride;
  launch_reduce_partials3(ws, nparts, stride, (int64_t)h * h, 1, dwre, nullptr, nullptr, ws2, 1.0f, s);
  launch_reduce_partials3(ws + (int64_t)h * h, nparts, stride, h, 2, dgamma, dbeta, nullptr, ws2, 1.0f, s);
}
