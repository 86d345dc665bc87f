// Pieces of the materialised attention path (scores in fp32 in a workspace), used by the FP32_PARITY
// mode and by the DeepViT / CaiT variants whose attention mixes information ACROSS heads:
//   softmax (vit.py:58,78), DeepViT re-attention head mix + LayerNorm over heads (deepvit.py:57-63,83-84),
//   CaiT talking heads (cait.py:97-98,123-125), context concat (cait.py:109-112), LayerScale VJP
//   (cait.py:47-48).  Score tensors are [b, h, nq, ld] fp32 with ld = round_up(nk, 4).
#include "kernels.h"

namespace {

constexpr int MAXH = 32;  // head-axis kernels keep one value per head in registers

// one wave per row; rows are short (nk <= a few hundred)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ sc, int64_t rows, int n, int64_t ld) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* r = sc + row * ld;
  float m = -INFINITY;
  for (int c = lane; c < n; c += 64) m = fmaxf(m, r[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < n; c += 64) { const float e = expf(r[c] - m); r[c] = e; s += e; }
  s = wave_sum(s);
  const float inv = 1.0f / s;
  for (int c = lane; c < n; c += 64) r[c] *= inv;
}
// dS = P * (dP - sum_j dP*P)   (in place on dP)
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ p, float* __restrict__ dp, int64_t rows, int n,
                                                               int64_t ld) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* pr = p + row * ld;
  float* dr = dp + row * ld;
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += pr[c] * dr[c];
  s = wave_sum(s);
  for (int c = lane; c < n; c += 64) dr[c] = pr[c] * (dr[c] - s);
}

// Short rows (n <= 128: the 64/65-key configurations): 16 lanes per row, four rows per wave -- a 65-key row on a whole wave leaves
// most lanes idle in its second pass.
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void softmax_rows16_kernel(float* __restrict__ sc, int64_t rows, int n, int64_t ld) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  const bool ok = row < rows;
  float* r = sc + (ok ? row : 0) * ld;
  float m = -INFINITY;
  if (ok) for (int c = l; c < n; c += 16) m = fmaxf(m, r[c]);
  m = group16_max(m);
  float s = 0.f;
  if (ok) for (int c = l; c < n; c += 16) { const float e = expf(r[c] - m); r[c] = e; s += e; }
  s = group16_sum(s);
  const float inv = 1.0f / s;
  if (ok) for (int c = l; c < n; c += 16) r[c] *= inv;
}
__global__ __launch_bounds__(256) void softmax_bwd_rows16_kernel(const float* __restrict__ p, float* __restrict__ dp, int64_t rows, int n,
                                                                 int64_t ld) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  const bool ok = row < rows;
  const float* pr = p + (ok ? row : 0) * ld;
  float* dr = dp + (ok ? row : 0) * ld;
  float s = 0.f;
  if (ok) for (int c = l; c < n; c += 16) s += pr[c] * dr[c];
  s = group16_sum(s);
  if (ok) for (int c = l; c < n; c += 16) dr[c] = pr[c] * (dr[c] - s);
}
// row statistics only: stats[row] = (max, 1 / sum exp(x - max))
__global__ __launch_bounds__(256) void softmax_stats16_kernel(const float* __restrict__ sc, float2* __restrict__ stats, int64_t rows, int n,
                                                              int64_t ld) {
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  const bool ok = row < rows;
  const float* r = sc + (ok ? row : 0) * ld;
  float m = -INFINITY;
  if (ok) for (int c = l; c < n; c += 16) m = fmaxf(m, r[c]);
  m = group16_max(m);
  float s = 0.f;
  if (ok) for (int c = l; c < n; c += 16) s += expf(r[c] - m);
  s = group16_sum(s);
  if (ok && l == 0) stats[row] = make_float2(m, 1.0f / s);
}

// DeepViT forward chain with one thread per (image, query, key) point (deepvit.py:80-84): softmax normalisation from the row
// statistics above, re-attention mix, LayerNorm over heads -- one read of the scores instead of three kernels' worth of passes.
// Writes the softmax back over the scores and the mixed values only when the backward needs them (`keep`).
template <int HT>
__global__ __launch_bounds__(256) void deepvit_point_fwd_kernel(float* __restrict__ s0, const float2* __restrict__ stats,
                                                                const float* __restrict__ w, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ mixed,
                                                                float* __restrict__ a2, int keep, int b, int nq, int64_t nvalid_per_row,
                                                                int64_t ld, float eps) {
  constexpr int h = HT;
  // the mixing matrix is read straight from global memory with wave-uniform addresses: the compiler turns that into scalar loads
  // and SGPR operands.  Staged through LDS it gets hoisted into 256 VGPRs per lane and the kernel runs at one wave per SIMD.
  const int64_t plane = (int64_t)nq * ld;
  const int64_t total = (int64_t)b * plane;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = e / plane, ij = e - bi * plane;
    if ((ij % ld) >= nvalid_per_row) continue;
    const int64_t i = ij / ld;
    float y[HT], v[HT];
#pragma unroll
    for (int hh = 0; hh < h; ++hh) {
      const float2 st = stats[(bi * h + hh) * nq + i];
      y[hh] = expf(s0[(bi * h + hh) * plane + ij] - st.x) * st.y;
      if (keep) s0[(bi * h + hh) * plane + ij] = y[hh];
    }
    float mu = 0.f;
#pragma unroll
    for (int gg = 0; gg < h; ++gg) {   // (column-wise on purpose: the row-wise form makes the scheduler hoist all 16 row loads and spill SGPRs)
      float a = 0.f;
#pragma unroll
      for (int hh = 0; hh < h; ++hh) a = fmaf(y[hh], w[hh * h + gg], a);
      v[gg] = a;
      mu += a;
    }
    mu /= (float)h;
    float var = 0.f;
#pragma unroll
    for (int gg = 0; gg < h; ++gg) var += (v[gg] - mu) * (v[gg] - mu);
    const float rs = rsqrtf(var / (float)h + eps);
#pragma unroll
    for (int gg = 0; gg < h; ++gg) a2[(bi * h + gg) * plane + ij] = (v[gg] - mu) * rs * gamma[gg] + beta[gg];
  }
}

// out[b,g,i,j] = sum_h in[b,h,i,j] W[h,g]    -- one thread per (b,i,j)
template <int HT>
__global__ __launch_bounds__(256) void headmix_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          float* __restrict__ out, int b, int h_rt, int64_t plane, int64_t nvalid_per_row,
                                                          int64_t ld) {
  const int h = HT ? HT : h_rt;   // compile-time head count keeps the per-point arrays in registers
  // W is read from global memory with wave-uniform addresses (scalar loads, SGPR operands); staged through LDS it was hoisted into
  // h*h VGPRs per lane and the kernel ran at one wave per SIMD
  const int64_t total = (int64_t)b * plane;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = e / plane, ij = e - bi * plane;
    if ((ij % ld) >= nvalid_per_row) continue;
    const float* ip = in + bi * h * plane + ij;
    const float* wr = w + opaque_zero();   // keeps the scalar weight loads inside the loop (see common.h)
    float* op = out + bi * h * plane + ij;
    float v[MAXH];
    _Pragma("unroll") for (int hh = 0; hh < h; ++hh) v[hh] = ip[(int64_t)hh * plane];
    // row hh of W at a time (16 contiguous scalars live), every output accumulates over hh in ascending order as before
    float acc[MAXH];
    _Pragma("unroll") for (int gg = 0; gg < h; ++gg) acc[gg] = 0.f;
    _Pragma("unroll") for (int hh = 0; hh < h; ++hh) {
      _Pragma("unroll") for (int gg = 0; gg < h; ++gg) acc[gg] = fmaf(v[hh], wr[hh * h + gg], acc[gg]);
    }
    _Pragma("unroll") for (int gg = 0; gg < h; ++gg) op[(int64_t)gg * plane] = acc[gg];
  }
}

// din[b,h,i,j] = sum_g dout[b,g,i,j] W[h,g];  per-block partial of dW[h,g] = sum in[h]*dout[g]
template <int HT>
__global__ __launch_bounds__(256) void headmix_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                          const float* __restrict__ w, float* __restrict__ din,
                                                          float* __restrict__ dw_partial, int b, int h_rt, int64_t plane,
                                                          int64_t nvalid_per_row, int64_t ld) {
  const int h = HT ? HT : h_rt;
  __shared__ float ws[MAXH * MAXH];
  extern __shared__ float pts[];   // [256][2h]: in[h], dout[h] of each point handled by this block iteration
  for (int i = threadIdx.x; i < h * h; i += blockDim.x) ws[i] = w[i];
  // dW accumulation.  Specialised head counts (multiples of 4): the h x h matrix is cut into 4x4 blocks; a thread owns ONE block
  // (16 accumulators) and every NB-th... point group: per point it reads 4 inputs + 4 output-gradients from LDS (two 16-B reads)
  // for 16 FMAs -- the LDS, not HBM, bounds this kernel, and this form moves 2.5x fewer LDS bytes than one float4 per 4 FMAs.
  // Runtime head counts: thread t owns pairs t, t+256, ... (h*h <= 1024).
  constexpr int HB = HT / 4;                                   // 4x4 blocks per side
  constexpr int NBLK = HB * HB > 0 ? HB * HB : 1;              // blocks of dW
  constexpr int PG = 256 / NBLK;                               // point groups (threads beyond NBLK*PG idle in the dW phase)
  constexpr bool TILED = HT >= 4 && HT % 4 == 0 && HT <= MAXH;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const int blk = threadIdx.x % NBLK, pg = threadIdx.x / NBLK;
  const int hb = (blk / (HB > 0 ? HB : 1)) * 4, gb = (blk % (HB > 0 ? HB : 1)) * 4;
  const int64_t total = (int64_t)b * plane;
  const int64_t span = (int64_t)gridDim.x * blockDim.x;
  const int64_t iters = (total + span - 1) / span;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t e = it * span + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = e < total;
    int64_t bi = 0, ij = 0;
    if (valid) { bi = e / plane; ij = e - bi * plane; valid = (ij % ld) < nvalid_per_row; }
    __syncthreads();
    float* mine = pts + (int64_t)threadIdx.x * 2 * h;
    if (valid) {
      float dv[MAXH];
      _Pragma("unroll") for (int gg = 0; gg < h; ++gg) { dv[gg] = dout[(bi * h + gg) * plane + ij]; mine[h + gg] = dv[gg]; }
      _Pragma("unroll") for (int hh = 0; hh < h; ++hh) {
        mine[hh] = in[(bi * h + hh) * plane + ij];
        float a = 0.f;
        _Pragma("unroll") for (int gg = 0; gg < h; ++gg) a = fmaf(dv[gg], ws[hh * h + gg], a);
        din[(bi * h + hh) * plane + ij] = a;
      }
    } else {
      for (int k = 0; k < 2 * h; ++k) mine[k] = 0.f;
    }
    __syncthreads();
    if (TILED) {
      if (pg < PG) {
        for (int p = pg; p < 256; p += PG) {
          const float4 x4 = *(const float4*)(pts + p * 2 * HT + hb);
          const float4 d4 = *(const float4*)(pts + p * 2 * HT + HT + gb);
          const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ds[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a * 4 + c] = fmaf(xs[a], ds[c], acc[a * 4 + c]);
        }
      }
    } else {
      for (int k = 0; k < 4; ++k) {
        const int pair = threadIdx.x + 256 * k;
        if (pair < h * h) {
          const int hh = pair / h, gg = pair - hh * h;
          float a = acc[k];
          for (int p = 0; p < 256; ++p) a = fmaf(pts[p * 2 * h + hh], pts[p * 2 * h + h + gg], a);
          acc[k] = a;
        }
      }
    }
  }
  if (TILED) {
    // combine the point groups in fixed order: partial[blk][hh][g] = sum over pg of this block's accumulators
    __syncthreads();
    float* red = pts;                                 // [PG][NBLK][16]
    if (pg < PG) {
#pragma unroll
      for (int k = 0; k < 16; ++k) red[(pg * NBLK + blk) * 16 + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < NBLK * 16) {
      const int bq = threadIdx.x / 16, k = threadIdx.x % 16;
      float t = 0.f;
      for (int q = 0; q < PG; ++q) t += red[(q * NBLK + bq) * 16 + k];
      const int hh = (bq / (HB > 0 ? HB : 1)) * 4 + k / 4, gg = (bq % (HB > 0 ? HB : 1)) * 4 + k % 4;
      dw_partial[(int64_t)blockIdx.x * HT * HT + hh * HT + gg] = t;
    }
    return;
  }
  for (int k = 0; k < 4; ++k) {
    const int pair = threadIdx.x + 256 * k;
    if (pair < h * h) dw_partial[(int64_t)blockIdx.x * h * h + pair] = acc[k];
  }
}

// Same VJP with the mixing-matrix gradient on the fp32 matrix pipe: dW[h][g] = sum over points of in[h] * dout[g] is a
// [16 x points] x [points x 16] product; each wave transposes the per-lane vectors of its 64 points into MFMA operand layout through
// a private LDS scratch and issues 16 v_mfma_f32_16x16x4_f32 (exact fp32 products).  The LDS point buffer of the kernel above moved
// ~1.3 KB per point; this one moves 160 B per point.  Head counts 4 / 8 / 12 / 16 (padded to 16).
constexpr int HMM_PITCH = 20;
template <int H>
__global__ __launch_bounds__(256) void headmix_bwd_mfma_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                               const float* __restrict__ w, float* __restrict__ din,
                                                               float* __restrict__ dw_partial, int b, int64_t plane, int64_t nvalid_per_row,
                                                               int64_t ld) {
  __shared__ __attribute__((aligned(16))) float scratch[4][2][64 * HMM_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xs = scratch[wave][0];
  float* ys = scratch[wave][1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int64_t total = (int64_t)b * plane;
  const int64_t span = (int64_t)gridDim.x * blockDim.x;
  const int64_t iters = (total + span - 1) / span;      // every wave runs the same number of iterations (MFMA needs all lanes)
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t e = it * span + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = e < total;
    int64_t bi = 0, ij = 0;
    if (valid) { bi = e / plane; ij = e - bi * plane; valid = (ij % ld) < nvalid_per_row; }
    float xv[H], dv[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      xv[k] = valid ? in[(bi * H + k) * plane + ij] : 0.f;
      dv[k] = valid ? dout[(bi * H + k) * plane + ij] : 0.f;
    }
    if (valid) {
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        float a = 0.f;
#pragma unroll
        for (int gg = 0; gg < H; ++gg) a = fmaf(dv[gg], w[hh * H + gg], a);
        din[(bi * H + hh) * plane + ij] = a;
      }
    }
    float* xr = xs + lane * HMM_PITCH;
    float* yr = ys + lane * HMM_PITCH;
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      *(float4*)(xr + k) = make_float4(k + 0 < H ? xv[k + 0 < H ? k + 0 : 0] : 0.f, k + 1 < H ? xv[k + 1 < H ? k + 1 : 0] : 0.f,
                                       k + 2 < H ? xv[k + 2 < H ? k + 2 : 0] : 0.f, k + 3 < H ? xv[k + 3 < H ? k + 3 : 0] : 0.f);
      *(float4*)(yr + k) = make_float4(k + 0 < H ? dv[k + 0 < H ? k + 0 : 0] : 0.f, k + 1 < H ? dv[k + 1 < H ? k + 1 : 0] : 0.f,
                                       k + 2 < H ? dv[k + 2 < H ? k + 2 : 0] : 0.f, k + 3 < H ? dv[k + 3 < H ? k + 3 : 0] : 0.f);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS traffic is in order
    const int m = lane & 15, kp = lane >> 4;
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const float a = xs[(4 * st + kp) * HMM_PITCH + m];
      const float bq = ys[(4 * st + kp) * HMM_PITCH + m];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float* pw = dw_partial + ((int64_t)blockIdx.x * 4 + wave) * H * H;
  const int g = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int hh = 4 * (lane >> 4) + r;
    if (hh < H && g < H) pw[hh * H + g] = acc[r];
  }
}

// DeepViT backward chain up to the softmax, one thread per point: LayerNorm-over-heads VJP -> re-attention mix VJP (dW on the fp32
// matrix pipe, see headmix_bwd_mfma_kernel).  d(A2) in `da` -> d(A0) written back; the softmax VJP follows as its own (row) kernel.
// Per-wave partials: [dW (H*H) | dgamma (H) | dbeta (H)].
template <int H>
__global__ __launch_bounds__(256) void deepvit_point_bwd_kernel(const float* __restrict__ a0,
                                                                float* __restrict__ da, const float* __restrict__ w,
                                                                const float* __restrict__ gamma, float* __restrict__ partial, int b,
                                                                int64_t plane, int64_t nvalid_per_row, int64_t ld, float eps) {
  __shared__ __attribute__((aligned(16))) float scratch[4][2][64 * HMM_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xs = scratch[wave][0];
  float* ys = scratch[wave][1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float ag[H], ab[H];
#pragma unroll
  for (int g = 0; g < H; ++g) { ag[g] = 0.f; ab[g] = 0.f; }
  const int64_t total = (int64_t)b * plane;
  const int64_t span = (int64_t)gridDim.x * blockDim.x;
  const int64_t iters = (total + span - 1) / span;      // uniform trip count: the MFMA needs every lane
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t e = it * span + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = e < total;
    int64_t bi = 0, ij = 0;
    if (valid) { bi = e / plane; ij = e - bi * plane; valid = (ij % ld) < nvalid_per_row; }
    float av[H], xh[H], dm[H];
    float mu = 0.f;
#pragma unroll
    for (int g = 0; g < H; ++g) av[g] = valid ? a0[(bi * H + g) * plane + ij] : 0.f;
    {
      // the mixed scores (deepvit.py:83) are recomputed from the softmax instead of being kept by the forward: 256 FMAs per point
      // against a [b, h, n, n] fp32 tensor written and read back; same FMA order as the forward kernels, i.e. the same bits
      const float* wm = w + opaque_zero();
#pragma unroll
      for (int g = 0; g < H; ++g) {
        float a = 0.f;
#pragma unroll
        for (int hh = 0; hh < H; ++hh) a = fmaf(av[hh], wm[hh * H + g], a);
        xh[g] = a;
        mu += a;
      }
    }
    mu /= (float)H;
    float var = 0.f;
#pragma unroll
    for (int g = 0; g < H; ++g) { xh[g] -= mu; var += xh[g] * xh[g]; }
    const float rs = rsqrtf(var / (float)H + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int g = 0; g < H; ++g) {                                    // same order as headnorm_bwd_kernel
      xh[g] *= rs;
      const float d = valid ? da[(bi * H + g) * plane + ij] : 0.f;
      ab[g] += d;
      ag[g] += d * xh[g];
      dm[g] = d * gamma[g];
      s1 += dm[g];
      s2 += dm[g] * xh[g];
    }
    s1 /= (float)H;
    s2 /= (float)H;
#pragma unroll
    for (int g = 0; g < H; ++g) dm[g] = valid ? rs * (dm[g] - s1 - xh[g] * s2) : 0.f;
    if (valid) {
      const float* wr = w + opaque_zero();
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < H; ++g) a = fmaf(dm[g], wr[hh * H + g], a);
        da[(bi * H + hh) * plane + ij] = a;
      }
    }
    float* xr = xs + lane * HMM_PITCH;
    float* yr = ys + lane * HMM_PITCH;
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      *(float4*)(xr + k) = make_float4(k + 0 < H ? av[k + 0 < H ? k + 0 : 0] : 0.f, k + 1 < H ? av[k + 1 < H ? k + 1 : 0] : 0.f,
                                       k + 2 < H ? av[k + 2 < H ? k + 2 : 0] : 0.f, k + 3 < H ? av[k + 3 < H ? k + 3 : 0] : 0.f);
      *(float4*)(yr + k) = make_float4(k + 0 < H ? dm[k + 0 < H ? k + 0 : 0] : 0.f, k + 1 < H ? dm[k + 1 < H ? k + 1 : 0] : 0.f,
                                       k + 2 < H ? dm[k + 2 < H ? k + 2 : 0] : 0.f, k + 3 < H ? dm[k + 3 < H ? k + 3 : 0] : 0.f);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int m = lane & 15, kp = lane >> 4;
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const float a = xs[(4 * st + kp) * HMM_PITCH + m];
      const float bq = ys[(4 * st + kp) * HMM_PITCH + m];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float* pw = partial + ((int64_t)blockIdx.x * 4 + wave) * (H * H + 2 * H);
  const int g16 = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int hh = 4 * (lane >> 4) + r;
    if (hh < H && g16 < H) pw[hh * H + g16] = acc[r];
  }
#pragma unroll
  for (int g = 0; g < H; ++g) {
    const float sg = wave_sum(ag[g]), sb = wave_sum(ab[g]);
    if (lane == 0) { pw[H * H + g] = sg; pw[H * H + H + g] = sb; }
  }
}

// LayerNorm over heads at every (b,i,j)  (deepvit.py:59-63)
template <int HT>
__global__ __launch_bounds__(256) void headnorm_fwd_kernel(const float* __restrict__ in, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ out, int b, int h_rt,
                                                           int64_t plane, int64_t nvalid_per_row, int64_t ld, float eps) {
  const int h = HT ? HT : h_rt;
  const int64_t total = (int64_t)b * plane;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = e / plane, ij = e - bi * plane;
    if ((ij % ld) >= nvalid_per_row) continue;
    float v[MAXH];
    float mu = 0.f;
    _Pragma("unroll") for (int hh = 0; hh < h; ++hh) { v[hh] = in[(bi * h + hh) * plane + ij]; mu += v[hh]; }
    mu /= (float)h;
    float var = 0.f;
    _Pragma("unroll") for (int hh = 0; hh < h; ++hh) var += (v[hh] - mu) * (v[hh] - mu);
    const float rs = rsqrtf(var / (float)h + eps);
    _Pragma("unroll") for (int hh = 0; hh < h; ++hh) out[(bi * h + hh) * plane + ij] = (v[hh] - mu) * rs * gamma[hh] + beta[hh];
  }
}
template <int HT>
__global__ __launch_bounds__(256) void headnorm_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                           const float* __restrict__ gamma, float* __restrict__ din,
                                                           float* __restrict__ partial, int b, int h_rt, int64_t plane,
                                                           int64_t nvalid_per_row, int64_t ld, float eps) {
  const int h = HT ? HT : h_rt;
  __shared__ float red[256];
  float ag[MAXH], ab[MAXH];
  _Pragma("unroll") for (int hh = 0; hh < h; ++hh) { ag[hh] = 0.f; ab[hh] = 0.f; }
  const int64_t total = (int64_t)b * plane;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = e / plane, ij = e - bi * plane;
    if ((ij % ld) >= nvalid_per_row) continue;
    float xh[MAXH], gg[MAXH];
    float mu = 0.f;
    _Pragma("unroll") for (int hh = 0; hh < h; ++hh) { xh[hh] = in[(bi * h + hh) * plane + ij]; mu += xh[hh]; }
    mu /= (float)h;
    float var = 0.f;
    _Pragma("unroll") for (int hh = 0; hh < h; ++hh) { xh[hh] -= mu; var += xh[hh] * xh[hh]; }
    const float rs = rsqrtf(var / (float)h + eps);
    float s1 = 0.f, s2 = 0.f;
    _Pragma("unroll") for (int hh = 0; hh < h; ++hh) {
      xh[hh] *= rs;
      const float d = dout[(bi * h + hh) * plane + ij];
      ab[hh] += d;
      ag[hh] += d * xh[hh];
      gg[hh] = d * gamma[hh];
      s1 += gg[hh];
      s2 += gg[hh] * xh[hh];
    }
    s1 /= (float)h;
    s2 /= (float)h;
    _Pragma("unroll") for (int hh = 0; hh < h; ++hh) din[(bi * h + hh) * plane + ij] = rs * (gg[hh] - s1 - xh[hh] * s2);
  }
  // fixed-order block reduction of the 2h accumulators
  _Pragma("unroll") for (int k = 0; k < 2 * h; ++k) {
    __syncthreads();
    red[threadIdx.x] = k < h ? ag[k] : ab[k - h];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * 2 * h + k] = red[0];
  }
}

template <typename TY, typename TC>
__global__ void concat_ctx_kernel(const TY* __restrict__ y, const float* __restrict__ context, TC* __restrict__ ctx, int b, int nq, int nc,
                                  int d) {
  const int64_t total = (int64_t)b * (nq + nc) * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % d);
    const int64_t r = e / d;
    const int64_t bi = r / (nq + nc);
    const int t = (int)(r - bi * (nq + nc));
    const float v = t < nq ? ldf<TY>(y + (bi * nq + t) * d + c) : context[(bi * nc + (t - nq)) * d + c];
    stf<TC>(ctx + e, v);
  }
}
template <typename T>
__global__ void split_ctx_bwd_kernel(const T* __restrict__ dctx, T* __restrict__ dy, float* __restrict__ dcontext, int b, int nq, int nc,
                                     int d) {
  const int64_t total = (int64_t)b * (nq + nc) * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % d);
    const int64_t r = e / d;
    const int64_t bi = r / (nq + nc);
    const int t = (int)(r - bi * (nq + nc));
    const float v = ldf<T>(dctx + e);
    if (t < nq) stf<T>(dy + (bi * nq + t) * d + c, v);
    else dcontext[(bi * nc + (t - nq)) * d + c] += v;
  }
}
template <typename T>
__global__ void add_T_kernel(T* __restrict__ a, const T* __restrict__ b2, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    stf<T>(a + e, ldf<T>(a + e) + ldf<T>(b2 + e));
}
// partial[chunk][c] = sum_{rows in chunk} g[r][c] * fx[r][c]; a wave owns 64 x 4 consecutive columns, the 4 waves of a block
// take interleaved rows (same shape as the bias column sums)
template <typename T>
__global__ __launch_bounds__(256) void scale_grad_kernel(const T* __restrict__ fx, int64_t ldf_, const float* __restrict__ g, int64_t ldg,
                                                         int rows, int d, float* __restrict__ partial, const float* __restrict__ scale,
                                                         T* __restrict__ out, int64_t ldo, int want_colsum) {
  // want_colsum (with out): the partial rows are [chunk][2 d] -- the second half holds the column sums of what was STORED in out (g * scale as T),
  // i.e. the bias gradient of the Dense layer in front of the LayerScale (cait.py:47-48 after fc2 / to_out): no separate pass over out
  __shared__ float red[4][256];
  __shared__ float redb[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int rows_per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto put = [&](int r, float4 v) {   // store g * scale as T and add what T holds to the column sums
    st4<T>(out + (int64_t)r * ldo + c, v);
    if (want_colsum) {
      bsum.x += (float)(T)v.x; bsum.y += (float)(T)v.y; bsum.z += (float)(T)v.z; bsum.w += (float)(T)v.w;
    }
  };
  // out != null (host: only when d % 4 == 0): the same pass also writes the gradient entering the branch, out = g * scale
  // (cait.py:47-48 VJP), instead of a second kernel reading g again
  if (c + 3 < d) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (out) sc = *(const float4*)(scale + c);
    int r = r0 + w;
    for (; r + 4 < r1; r += 8) {
      const float4 g0 = *(const float4*)(g + (int64_t)r * ldg + c), g1 = *(const float4*)(g + (int64_t)(r + 4) * ldg + c);
      const float4 f0 = ld4<T>(fx + (int64_t)r * ldf_ + c), f1 = ld4<T>(fx + (int64_t)(r + 4) * ldf_ + c);
      a.x += g0.x * f0.x + g1.x * f1.x; a.y += g0.y * f0.y + g1.y * f1.y; a.z += g0.z * f0.z + g1.z * f1.z; a.w += g0.w * f0.w + g1.w * f1.w;
      if (out) {
        put(r, make_float4(g0.x * sc.x, g0.y * sc.y, g0.z * sc.z, g0.w * sc.w));
        put(r + 4, make_float4(g1.x * sc.x, g1.y * sc.y, g1.z * sc.z, g1.w * sc.w));
      }
    }
    for (; r < r1; r += 4) {
      const float4 g0 = *(const float4*)(g + (int64_t)r * ldg + c);
      const float4 f0 = ld4<T>(fx + (int64_t)r * ldf_ + c);
      a.x += g0.x * f0.x; a.y += g0.y * f0.y; a.z += g0.z * f0.z; a.w += g0.w * f0.w;
      if (out) put(r, make_float4(g0.x * sc.x, g0.y * sc.y, g0.z * sc.z, g0.w * sc.w));
    }
  } else if (c < d) {
    float* ap = (float*)&a;
    for (int r = r0 + w; r < r1; r += 4)
      for (int i = 0; i < 4 && c + i < d; ++i) ap[i] += g[(int64_t)r * ldg + c + i] * ldf<T>(fx + (int64_t)r * ldf_ + c + i);
  }
  *(float4*)&red[w][lane * 4] = a;
  if (want_colsum) *(float4*)&redb[w][lane * 4] = bsum;
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  const int64_t prow = want_colsum ? 2 * (int64_t)d : (int64_t)d;
  if (cc < d) {
    partial[(int64_t)blockIdx.y * prow + cc] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (want_colsum) partial[(int64_t)blockIdx.y * prow + d + cc] = (redb[0][threadIdx.x] + redb[1][threadIdx.x]) + (redb[2][threadIdx.x] + redb[3][threadIdx.x]);
  }
}
template <typename TO>
__global__ void mul_scale_kernel(const float* __restrict__ g, int64_t ldg, const float* __restrict__ scale, TO* __restrict__ out,
                                 int64_t ldo, int rows, int d) {
  const int64_t total = (int64_t)rows * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / d;
    const int c = (int)(e - r * d);
    stf<TO>(out + r * ldo + c, g[r * ldg + c] * scale[c]);
  }
}
__global__ void broadcast_rows_kernel(const float* __restrict__ src, int d, float* __restrict__ dst, int rows) {
  const int64_t total = (int64_t)rows * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    dst[e] = src[e % d];
}

#define VITX_H_DISPATCH(h, CALL)                                                              \
  do {                                                                                       \
    if ((h) == 16) { CALL(16); } else if ((h) == 12) { CALL(12); } else if ((h) == 8) { CALL(8); } \
    else if ((h) == 4) { CALL(4); } else { CALL(0); }                                         \
  } while (0)

inline int grid_for(int64_t total, int block = 256) { return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(total, block), 256 * 8)); }
constexpr int HM_BLOCKS = 256;
constexpr int HMM_BLOCKS = 2048;   // headmix_bwd_mfma_kernel: 4 resident workgroups per CU x 256 CUs x 2
constexpr int SG_CHUNKS = 512;

}  // namespace

void launch_softmax_rows(float* sc, int64_t rows, int n, int64_t ld, hipStream_t s) {
  if (rows == 0) return;
  if (n <= 128) hipLaunchKernelGGL(softmax_rows16_kernel, dim3((unsigned)ceil_div(rows, 16)), dim3(256), 0, s, sc, rows, n, ld);
  else hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, sc, rows, n, ld);
}
void launch_softmax_bwd_rows(const float* p, float* dp, int64_t rows, int n, int64_t ld, hipStream_t s) {
  if (rows == 0) return;
  if (n <= 128) hipLaunchKernelGGL(softmax_bwd_rows16_kernel, dim3((unsigned)ceil_div(rows, 16)), dim3(256), 0, s, p, dp, rows, n, ld);
  else hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, p, dp, rows, n, ld);
}
// DeepViT backward chain: fused point kernel (LayerNorm-over-heads VJP + mix VJP), then the row softmax VJP
int64_t deepvit_point_bwd_ws_elems(int h) { return (int64_t)(HMM_BLOCKS * 4 + 40) * (h * h + 2 * h); }
void launch_deepvit_point_bwd(const float* a0, float* da_inout, const float* w, const float* gamma, float* ws, float* dw,
                              float* dgamma, float* dbeta, int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s) {
  const int64_t plane = (int64_t)nq * ld;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(HMM_BLOCKS, ceil_div((int64_t)b * plane, 256)));
#define CALLB(HT) hipLaunchKernelGGL(deepvit_point_bwd_kernel<HT>, dim3(nb), dim3(256), 0, s, a0, da_inout, w, gamma, ws, b, plane, (int64_t)nk, ld, eps)
  if (h == 16) { CALLB(16); } else if (h == 12) { CALLB(12); } else if (h == 8) { CALLB(8); } else { CALLB(4); }
#undef CALLB
  const int nparts = nb * 4;
  const int64_t stride = (int64_t)h * h + 2 * h;
  float* ws2 = ws + (int64_t)nparts * stride;
  launch_reduce_partials3(ws, nparts, stride, (int64_t)h * h, 1, dw, nullptr, nullptr, ws2, 1.0f, s);
  launch_reduce_partials3(ws + (int64_t)h * h, nparts, stride, h, 2, dgamma, dbeta, nullptr, ws2, 1.0f, s);
  launch_softmax_bwd_rows(a0, da_inout, (int64_t)b * h * nq, nk, ld, s);
}
// DeepViT forward chain in two launches: row statistics, then one fused point kernel (see deepvit_point_fwd_kernel)
bool deepvit_point_fwd_supported(int h, int nk) { return (h == 4 || h == 8 || h == 12 || h == 16) && nk <= 128; }
int64_t deepvit_point_ws_elems(int b, int h, int nq) { return 2 * (int64_t)b * h * nq; }
void launch_deepvit_point_fwd(float* s0, float* stats_ws, const float* w, const float* gamma, const float* beta, float* mixed, float* a2,
                              int keep, int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s) {
  const int64_t rows = (int64_t)b * h * nq;
  if (rows == 0) return;
  hipLaunchKernelGGL(softmax_stats16_kernel, dim3((unsigned)ceil_div(rows, 16)), dim3(256), 0, s, s0, (float2*)stats_ws, rows, nk, ld);
  const int64_t plane = (int64_t)nq * ld;
#define CALLP(HT) hipLaunchKernelGGL(deepvit_point_fwd_kernel<HT>, dim3(grid_for((int64_t)b * plane)), dim3(256), 0, s, s0, (const float2*)stats_ws, w, gamma, beta, mixed, a2, keep, b, nq, (int64_t)nk, ld, eps)
  if (h == 16) { CALLP(16); } else if (h == 12) { CALLP(12); } else if (h == 8) { CALLP(8); } else { CALLP(4); }
#undef CALLP
}
void launch_headmix_fwd(const float* in, const float* w, float* out, int b, int h, int nq, int nk, int64_t ld, hipStream_t s) {
  const int64_t plane = (int64_t)nq * ld;
#define CALL(HT) hipLaunchKernelGGL(headmix_fwd_kernel<HT>, dim3(grid_for((int64_t)b * plane)), dim3(256), 0, s, in, w, out, b, h, plane, (int64_t)nk, ld)
  VITX_H_DISPATCH(h, CALL);
#undef CALL
}
int64_t headmix_ws_elems(int b, int h, int nq, int nk) { return (int64_t)(HMM_BLOCKS * 4 + 40) * h * h; }
void launch_headmix_bwd(const float* in, const float* dout, const float* w, float* din, float* dw_partial_ws, float* dw, int b, int h,
                        int nq, int nk, int64_t ld, hipStream_t s) {
  const int64_t plane = (int64_t)nq * ld;
  const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(HM_BLOCKS, ceil_div((int64_t)b * plane, 256)));
  const size_t shm = (size_t)std::max(256 * 2 * h, 4096) * sizeof(float);   // points [256][2h]; later the [PG][blocks][16] combine buffer
  if (h == 4 || h == 8 || h == 12 || h == 16) {   // mixing-matrix gradient on the fp32 matrix pipe, one partial per wave
    const int nb2 = (int)std::max<int64_t>(1, std::min<int64_t>(HMM_BLOCKS, ceil_div((int64_t)b * plane, 256)));
#define CALLM(HT) hipLaunchKernelGGL(headmix_bwd_mfma_kernel<HT>, dim3(nb2), dim3(256), 0, s, in, dout, w, din, dw_partial_ws, b, plane, (int64_t)nk, ld)
    if (h == 16) { CALLM(16); } else if (h == 12) { CALLM(12); } else if (h == 8) { CALLM(8); } else { CALLM(4); }
#undef CALLM
    const int nparts = nb2 * 4;
    launch_reduce_partials3(dw_partial_ws, nparts, (int64_t)h * h, (int64_t)h * h, 1, dw, nullptr, nullptr, dw_partial_ws + (int64_t)nparts * h * h, 1.0f, s);
    return;
  }
#define CALL(HT) hipLaunchKernelGGL(headmix_bwd_kernel<HT>, dim3(nblk), dim3(256), shm, s, in, dout, w, din, dw_partial_ws, b, h, plane, (int64_t)nk, ld)
  VITX_H_DISPATCH(h, CALL);
#undef CALL
  launch_reduce_partials(dw_partial_ws, nblk, (int64_t)h * h, (int64_t)h * h, dw, 1.0f, s);
}
void launch_headnorm_fwd(const float* in, const float* gamma, const float* beta, float* out, int b, int h, int nq, int nk, int64_t ld,
                         float eps, hipStream_t s) {
  const int64_t plane = (int64_t)nq * ld;
#define CALL(HT) hipLaunchKernelGGL(headnorm_fwd_kernel<HT>, dim3(grid_for((int64_t)b * plane)), dim3(256), 0, s, in, gamma, beta, out, b, h, plane, (int64_t)nk, ld, eps)
  VITX_H_DISPATCH(h, CALL);
#undef CALL
}
void launch_headnorm_bwd(const float* in, const float* dout, const float* gamma, float* din, float* partial_ws, float* dgamma, float* dbeta,
                         int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s) {
  const int64_t plane = (int64_t)nq * ld;
  const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(HM_BLOCKS, ceil_div((int64_t)b * plane, 256)));
#define CALL(HT) hipLaunchKernelGGL(headnorm_bwd_kernel<HT>, dim3(nblk), dim3(256), 0, s, in, dout, gamma, din, partial_ws, b, h, plane, (int64_t)nk, ld, eps)
  VITX_H_DISPATCH(h, CALL);
#undef CALL
  launch_reduce_partials(partial_ws, nblk, (int64_t)2 * h, h, dgamma, 1.0f, s);
  launch_reduce_partials(partial_ws + h, nblk, (int64_t)2 * h, h, dbeta, 1.0f, s);
}
void launch_concat_ctx(const void* y, int y_bf16, const float* context, void* ctx, int ctx_bf16, int b, int nq, int nc, int d, hipStream_t s) {
  const int64_t total = (int64_t)b * (nq + nc) * d;
  if (y_bf16 && ctx_bf16) hipLaunchKernelGGL((concat_ctx_kernel<bf16_t, bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)y, context, (bf16_t*)ctx, b, nq, nc, d);
  else hipLaunchKernelGGL((concat_ctx_kernel<float, float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)y, context, (float*)ctx, b, nq, nc, d);
}
void launch_split_ctx_bwd(const void* dctx, int is_bf16, void* dy, float* dcontext, int b, int nq, int nc, int d, hipStream_t s) {
  const int64_t total = (int64_t)b * (nq + nc) * d;
  if (is_bf16) hipLaunchKernelGGL(split_ctx_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dctx, (bf16_t*)dy, dcontext, b, nq, nc, d);
  else hipLaunchKernelGGL(split_ctx_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)dctx, (float*)dy, dcontext, b, nq, nc, d);
}
void launch_add_T(void* a, const void* b2, int is_bf16, int64_t n, hipStream_t s) {
  if (n == 0) return;
  if (is_bf16) hipLaunchKernelGGL(add_T_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, (bf16_t*)a, (const bf16_t*)b2, n);
  else hipLaunchKernelGGL(add_T_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (float*)a, (const float*)b2, n);
}
void launch_scale_grad(const void* fx, int is_bf16, int64_t ldf_, const float* g, int64_t ldg, int rows, int d, float* partial_ws,
                       float* dscale, hipStream_t s, const float* scale, void* out, int64_t ldo, float* dbias) {
  static const int blocks = [] { const char* v = vitx_env("VITX_SG_BLOCKS"); return v ? atoi(v) : 2048; }();   // (experiment: blocks per launch)
  const int cblocks = (int)ceil_div(d, 256);
  const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)SG_CHUNKS, ceil_div(rows, 16), ceil_div(blocks, cblocks)}));
  const int want = (dbias != nullptr && out != nullptr) ? 1 : 0;   // partial_ws holds SG_CHUNKS x 2 d sums + 64 d of second-level scratch
  dim3 grid((unsigned)cblocks, chunks), block(256);
  if (is_bf16) hipLaunchKernelGGL(scale_grad_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)fx, ldf_, g, ldg, rows, d, partial_ws, scale, (bf16_t*)out, ldo, want);
  else hipLaunchKernelGGL(scale_grad_kernel<float>, grid, block, 0, s, (const float*)fx, ldf_, g, ldg, rows, d, partial_ws, scale, (float*)out, ldo, want);
  if (want) launch_reduce_partials3(partial_ws, chunks, 2 * (int64_t)d, d, 2, dscale, dbias, nullptr, partial_ws + (int64_t)SG_CHUNKS * 2 * d, 1.0f, s);
  else launch_reduce_partials3(partial_ws, chunks, d, d, 1, dscale, nullptr, nullptr, partial_ws + (int64_t)SG_CHUNKS * d, 1.0f, s);
}
void launch_mul_scale(const float* g, int64_t ldg, const float* scale, void* out, int out_bf16, int64_t ldo, int rows, int d, hipStream_t s) {
  const int64_t total = (int64_t)rows * d;
  if (total == 0) return;
  if (out_bf16) hipLaunchKernelGGL(mul_scale_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, g, ldg, scale, (bf16_t*)out, ldo, rows, d);
  else hipLaunchKernelGGL(mul_scale_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, g, ldg, scale, (float*)out, ldo, rows, d);
}
void launch_broadcast_rows(const float* src, int d, float* dst, int rows, hipStream_t s) {
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3(grid_for((int64_t)rows * d)), dim3(256), 0, s, src, d, dst, rows);
}
