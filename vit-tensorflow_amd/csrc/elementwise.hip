// HBM-bound kernels of the ViT path: patch unfold, LayerNorm fwd/bwd, bias-gradient column sums,
// operand conversion/transposes, cls/pos rows, pooling, loss gradient.  One wave (64 lanes) per
// token row with float4 accesses wherever a row is reduced; reductions are two-stage and
// fixed-order (deterministic, no atomics).
#include "kernels.h"

namespace {

// ------------------------------------------------------------------ patch unfold (vit.py:142)
// out[(b*Hp + hi)*Wp + wi][(r*pw + s)*C + c] = img[b][hi*ph + r][wi*pw + s][c]   -- pure indexing.
// Consecutive threads walk the (s,c) run, which is contiguous in both img and out.
template <typename TO>
__global__ void unfold_kernel(const float* __restrict__ img, TO* __restrict__ out, int b, int H, int W, int C, int ph, int pw,
                              int64_t ldo) {
  const int Hp = H / ph, Wp = W / pw;
  const int pd = ph * pw * C;
  const int run = pw * C;
  const int64_t total = (int64_t)b * Hp * Wp * ldo;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / ldo;
    const int f = (int)(e - row * ldo);
    float v = 0.f;
    if (f < pd) {
      const int r = f / run, sc = f - r * run;
      const int wi = (int)(row % Wp);
      const int64_t t = row / Wp;
      const int hi = (int)(t % Hp);
      const int64_t bi = t / Hp;
      v = img[((bi * H + (int64_t)hi * ph + r) * W + (int64_t)wi * pw) * C + sc];
    }
    stf<TO>(out + e, v);
  }
}

// four consecutive features per thread (16-B load, 8/16-B store); legal when the contiguous (s,c) run and the row pitch are
// multiples of 4 and the image base / row starts are 16-B aligned -- bit-exact like the scalar form (pure indexing)
template <typename TO>
__global__ void unfold4_kernel(const float* __restrict__ img, TO* __restrict__ out, int b, int H, int W, int C, int ph, int pw, int ldo) {
  const int Hp = H / ph, Wp = W / pw;
  const int pd = ph * pw * C, run = pw * C, ldo4 = ldo >> 2;
  const int64_t total = (int64_t)b * Hp * Wp * ldo4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / ldo4;
    const int f = (int)(e - row * ldo4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < pd) {
      const int r = f / run, sc = f - r * run;
      const int wi = (int)(row % Wp);
      const int64_t t = row / Wp;
      const int hi = (int)(t % Hp);
      const int64_t bi = t / Hp;
      v = *(const float4*)(img + ((bi * H + (int64_t)hi * ph + r) * W + (int64_t)wi * pw) * C + sc);
    }
    st4<TO>(out + row * ldo + f, v);
  }
}

// dimg[b][hi*ph+r][wi*pw+s][c] = dpatches[(b,hi,wi)][(r,s,c)]   (inverse of the unfold; bijective)
__global__ void fold_kernel(const float* __restrict__ dp, int64_t ld, float* __restrict__ dimg, int b, int H, int W, int C, int ph,
                            int pw) {
  const int Hp = H / ph, Wp = W / pw;
  const int pd = ph * pw * C, run = pw * C;
  const int64_t total = (int64_t)b * Hp * Wp * pd;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / pd;
    const int f = (int)(e - row * pd);
    const int r = f / run, sc = f - r * run;
    const int wi = (int)(row % Wp);
    const int64_t t = row / Wp;
    const int hi = (int)(t % Hp);
    const int64_t bi = t / Hp;
    dimg[((bi * H + (int64_t)hi * ph + r) * W + (int64_t)wi * pw) * C + sc] = dp[row * ld + f];
  }
}

// x[b*ntok + 0][:] = cls + pos[0]   (vit.py:163-165)
__global__ void cls_pos_row_kernel(float* x, const float* cls, const float* pos, int b, int ntok, int d, int64_t ldx) {
  const int64_t total = (int64_t)b * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = e / d;
    const int c = (int)(e - bi * d);
    x[bi * ntok * ldx + c] = cls[c] + pos[c];
  }
}

// ------------------------------------------------------------------ LayerNorm (vit.py:18,22,155; Keras eps 1e-3, biased var)
template <typename TO, int VPL>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, TO* __restrict__ y, int64_t ldy,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int rows, int d,
                                                            float eps) {
  const int row = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * ldx;
  float4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < d) { v[i] = *(const float4*)(xr + c); s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
    else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < d) {
      const float a = v[i].x - mu, b2 = v[i].y - mu, c2 = v[i].z - mu, d2 = v[i].w - mu;
      q += (a * a + b2 * b2) + (c2 * c2 + d2 * d2);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  TO* yr = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < d) {
      const float4 g = *(const float4*)(gamma + c), bb = *(const float4*)(beta + c);
      st4<TO>(yr + c, make_float4((v[i].x - mu) * rs * g.x + bb.x, (v[i].y - mu) * rs * g.y + bb.y,
                                  (v[i].z - mu) * rs * g.z + bb.z, (v[i].w - mu) * rs * g.w + bb.w));
    }
  }
}

// Any-width forms (d % 4 != 0, or rows that do not start 16-B aligned: T2T-ViT's 147- and 1323-wide token transformers, t2t.py:62-70):
// one wave per row, one element per lane and step.  Same arithmetic as the vector kernels (two-pass statistics, biased variance).
template <typename TO>
__global__ __launch_bounds__(256) void layernorm_fwd_any_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, TO* __restrict__ y, int64_t ldy,
                                                                float* __restrict__ mean, float* __restrict__ rstd, int rows, int d, float eps) {
  const int row = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * ldx;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += xr[c];
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int c = lane; c < d; c += 64) { const float a = xr[c] - mu; q += a * a; }
  const float rs = rsqrtf(wave_sum(q) / (float)d + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  TO* yr = y + (int64_t)row * ldy;
  for (int c = lane; c < d; c += 64) stf<TO>(yr + c, (xr[c] - mu) * rs * gamma[c] + beta[c]);
}
// dx (+ g_in) -> g_out / g_lp per row; the column reductions (dgamma, dbeta, column sums of g_in) are a second, column-parallel kernel
template <typename TD, typename TL>
__global__ __launch_bounds__(256) void layernorm_bwd_any_rows_kernel(const TD* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                     const float* __restrict__ gamma, const float* g_in, int64_t ldgi, float* g_out,
                                                                     int64_t ldgo, TL* g_lp, int64_t ldglp, int rows, int d) {
  const int row = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float mu = mean[row], rs = rstd[row];
  const float* xr = x + (int64_t)row * ldx;
  const TD* dr = dy + (int64_t)row * lddy;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float gg = ldf<TD>(dr + c) * gamma[c];
    s1 += gg;
    s2 += gg * ((xr[c] - mu) * rs);
  }
  const float invd = 1.0f / (float)d;
  s1 = wave_sum(s1) * invd;
  s2 = wave_sum(s2) * invd;
  for (int c = lane; c < d; c += 64) {
    const float xh = (xr[c] - mu) * rs;
    float dx = rs * (ldf<TD>(dr + c) * gamma[c] - s1 - xh * s2);
    if (g_in) dx += g_in[(int64_t)row * ldgi + c];
    g_out[(int64_t)row * ldgo + c] = dx;
    if (g_lp) stf<TL>(g_lp + (int64_t)row * ldglp + c, dx);
  }
}
// partial[chunk][0][c] = sum_rows dy*xhat, [1] = sum_rows dy, [2] = sum_rows g_in (rows of the chunk in ascending order: fixed order)
template <typename TD>
__global__ __launch_bounds__(256) void layernorm_bwd_any_cols_kernel(const TD* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                     const float* g_in, int64_t ldgi, float* __restrict__ partial, int rows, int d,
                                                                     int want_gsum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  const int per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float ag = 0.f, ab = 0.f, as = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float dv = ldf<TD>(dy + (int64_t)r * lddy + c);
    ab += dv;
    ag += dv * ((x[(int64_t)r * ldx + c] - mean[r]) * rstd[r]);
    if (want_gsum) as += g_in[(int64_t)r * ldgi + c];
  }
  float* p = partial + (int64_t)blockIdx.y * 3 * d;
  p[c] = ag; p[d + c] = ab; p[2 * d + c] = as;
}

#ifndef VITX_LNB_FUSE_WAVES
#define VITX_LNB_FUSE_WAVES 3
#endif
constexpr int LNB_BLOCKS = 512;
constexpr int LNB_BLOCKS_MAX = 1024;   // partial rows of the co-resident form (4-wave blocks: twice the blocks for the same waves in flight)
constexpr int LNB_THREADS = 512;   // 8 waves per block: 4096 waves in flight with only 512 partial rows to reduce

// dx = r * (g*gamma - mean_d(g*gamma) - xhat * mean_d(g*gamma*xhat));  dgamma += g*xhat; dbeta += g
// VPL = 4 (d = 1024) lands on 134 VGPRs by itself = 3 waves per SIMD = ONE 8-wave block per CU where d = 768 runs two; asking for
// 4 waves per SIMD caps it at 128 (ViT-L / DeepViT / CaiT: 3.9 -> 5.5 TB/s)
// FUSE (round 5; CaiT, cait.py:47-48): the residual gradient this pass produces is what the NEXT branch of the backward chain -- the one whose output
// f = Dense(...) was scaled by a LayerScale vector `nscale` and added to the stream -- needs for its LayerScale VJP: g_lp receives g * nscale (the
// gradient entering that branch's last Dense, in T), and two more partial rows hold column sums of g (x nscale = that Dense's bias gradient) and of
// g * f (= d nscale).  Replaces a pass of its own over g (fp32), f and the branch gradient (134 MB per branch at cfg5) by one more read of f here.
// CO (round 6): the form that fits BESIDE a resident weight-gradient GEMM.  A GEMM workgroup holds 8 waves x 208 VGPRs = 416 of a SIMD's 512 and
// 128 KiB of the CU's 160 KiB LDS: what is left is ONE wave of <= 96 VGPRs per SIMD and 32 KiB.  4 waves per block (one per SIMD), registers capped at
// 96 (launch bound 5 waves per SIMD), 12 KiB of LDS at d = 768: the dispatcher places such a block on a CU whose GEMM workgroup is mid K-slice, and the
// HBM-bound pass runs under the MFMA-bound one instead of waiting for it (tools/probe_coresident.hip; section 5 of DESIGN.md, round 6).
template <typename TD, typename TL, int VPL, bool FUSE, bool CO = false>
__global__ __launch_bounds__(CO ? 256 : LNB_THREADS, CO ? 5 : (VPL == 4 ? (FUSE ? VITX_LNB_FUSE_WAVES : 4) : 1)) void layernorm_bwd_kernel(const TD* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                            int64_t ldx, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* g_in, int64_t ldgi,
                                                            float* g_out, int64_t ldgo, TL* g_lp, int64_t ldglp,
                                                            float* __restrict__ partial, int rows, int d, int want_gsum,
                                                            const TL* __restrict__ nf, int64_t ldnf, const float* __restrict__ nscale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  constexpr int NSEG = FUSE ? 4 : 3;
  float4 ag[VPL], ab[VPL], gm[VPL], as[VPL];   // as: column sums of g_in (= bias gradient of the branch's last Dense); FUSE: of g_out
  float4 asc[FUSE ? VPL : 1];                  // FUSE: column sums of g_out * f
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    as[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FUSE) asc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (lane + 64 * i) * 4;
    gm[i] = (!FUSE && !CO && c < d) ? *(const float4*)(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invd = 1.0f / (float)d;
  for (int row = blockIdx.x * nw + wib; row < rows; row += gridDim.x * nw) {
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + (int64_t)row * ldx;
    const TD* dr = dy + (int64_t)row * lddy;
    float4 xh[VPL], gg[VPL];
    constexpr bool PRE = FUSE;   // (requesting g_in early in the plain form as well: 2.74 vs 2.72 ms per ViT-B/16 step, neutral -- profiles/r5/ab_layernorm_vjp_prefetch_r5q_neutral.log)
    float4 gip[PRE ? VPL : 1], fvp[FUSE ? VPL : 1];   // g_in (and f) of this row, requested with x and dy (not behind the two wave reductions)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < d) {
        const float4 xv = *(const float4*)(xr + c);
        const float4 dv = ld4<TD>(dr + c);
        if (PRE) gip[i] = g_in ? *(const float4*)(g_in + (int64_t)row * ldgi + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (FUSE) fvp[i] = ld4<TL>(nf + (int64_t)row * ldnf + c);
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
        ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
        const float4 gmv = (FUSE || CO) ? *(const float4*)(gamma + c) : gm[i];   // FUSE / CO: gamma from L1 per row (the registers the extra accumulators / the 96-register cap need)
        gg[i] = make_float4(dv.x * gmv.x, dv.y * gmv.y, dv.z * gmv.z, dv.w * gmv.w);
        s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
        s2 += (gg[i].x * xh[i].x + gg[i].y * xh[i].y) + (gg[i].z * xh[i].z + gg[i].w * xh[i].w);
      }
    }
    s1 = wave_sum(s1) * invd;
    s2 = wave_sum(s2) * invd;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < d) {
        float4 dx = make_float4(rs * (gg[i].x - s1 - xh[i].x * s2), rs * (gg[i].y - s1 - xh[i].y * s2),
                                rs * (gg[i].z - s1 - xh[i].z * s2), rs * (gg[i].w - s1 - xh[i].w * s2));
        if (PRE) {
          if (!FUSE) { as[i].x += gip[i].x; as[i].y += gip[i].y; as[i].z += gip[i].z; as[i].w += gip[i].w; }
          dx.x += gip[i].x; dx.y += gip[i].y; dx.z += gip[i].z; dx.w += gip[i].w;
        } else if (g_in) {
          const float4 gi = *(const float4*)(g_in + (int64_t)row * ldgi + c);
          as[i].x += gi.x; as[i].y += gi.y; as[i].z += gi.z; as[i].w += gi.w;
          dx.x += gi.x; dx.y += gi.y; dx.z += gi.z; dx.w += gi.w;
        }
        *(float4*)(g_out + (int64_t)row * ldgo + c) = dx;
        if (FUSE) {
          const float4 fv = fvp[i];
          const float4 sc = *(const float4*)(nscale + c);          // (4 KiB, L1-resident: not worth 16 registers at d = 1024)
          as[i].x += dx.x; as[i].y += dx.y; as[i].z += dx.z; as[i].w += dx.w;
          asc[i].x += dx.x * fv.x; asc[i].y += dx.y * fv.y; asc[i].z += dx.z * fv.z; asc[i].w += dx.w * fv.w;
          st4<TL>(g_lp + (int64_t)row * ldglp + c, make_float4(dx.x * sc.x, dx.y * sc.y, dx.z * sc.z, dx.w * sc.w));
        } else if (g_lp) {
          st4<TL>(g_lp + (int64_t)row * ldglp + c, dx);
        }
      }
    }
  }
  // fixed-order combine of the block's waves: dgamma, dbeta (, column sums of g_in), through LDS [nw][d]
  // (three explicit passes: selecting the accumulator array with a run-time pass index made the compiler keep all three arrays in
  //  scratch memory for the whole kernel -- 18 KB of scratch traffic per row next to 12 KB of useful HBM traffic)
  auto combine = [&](const auto& acc, int pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < d) *(float4*)(lds + (int64_t)wib * d + c) = acc[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
      float a = 0.f;
      for (int w = 0; w < nw; ++w) a += lds[(int64_t)w * d + c];
      partial[((int64_t)blockIdx.x * NSEG + pass) * d + c] = a;
    }
  };
  combine(ag, 0);
  combine(ab, 1);
  if (FUSE) { combine(as, 2); combine(asc, 3); }
  else if (want_gsum) combine(as, 2);
}

// out[c] = alpha * sum_p partial[p*stride + c]; 64 columns x 4 part-groups per block, fixed order.
// blockIdx.y selects a chunk of parts (two-level reduction when there are many parts and few columns);
// columns [0,seg) go to out0, [seg,2seg) to out1, [2seg,3seg) to out2 (seg = n when only out0 is used).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int nparts, int64_t stride, int64_t n,
                                                              float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2,
                                                              int64_t seg, int64_t out_chunk_stride, float alpha) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const int per = (nparts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  float a0 = 0.f, a1 = 0.f;
  if (c < n) {
    int p = p0 + pg;
    for (; p + 4 < p1; p += 8) { a0 += partial[(int64_t)p * stride + c]; a1 += partial[(int64_t)(p + 4) * stride + c]; }
    for (; p < p1; p += 4) a0 += partial[(int64_t)p * stride + c];
  }
  red[pg][lane] = a0 + a1;
  __syncthreads();
  if (pg == 0 && c < n) {
    const float v = alpha * ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
    if (out_chunk_stride) { out0[(int64_t)blockIdx.y * out_chunk_stride + c] = v; return; }
    const int64_t which = c / seg, cc = c - which * seg;
    (which == 0 ? out0 : (which == 1 ? out1 : out2))[cc] = v;
  }
}

// Single-output, single-level case on 16-B friendly operands (the split-K slices of a weight gradient: 7-28 parts of up to 2.4 M
// columns): four columns per lane, four parts in flight, parts summed in index order ((p0 + p1) + (p2 + p3) per group of four).
__global__ __launch_bounds__(256) void reduce_partials4_kernel(const float4* __restrict__ partial, int nparts, int64_t stride4, int64_t n4,
                                                               float4* __restrict__ out, float alpha) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n4) return;
  const float4* src = partial + c;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = 0;
  for (; p + 4 <= nparts; p += 4) {
    const float4 v0 = src[(int64_t)p * stride4], v1 = src[(int64_t)(p + 1) * stride4], v2 = src[(int64_t)(p + 2) * stride4],
                 v3 = src[(int64_t)(p + 3) * stride4];
    a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
    a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
  }
  for (; p < nparts; ++p) { const float4 v = src[(int64_t)p * stride4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  out[c] = make_float4(a.x * alpha, a.y * alpha, a.z * alpha, a.w * alpha);
}

// ------------------------------------------------------------------ column sums (Dense bias gradients: db = sum_rows dY)
constexpr int CS_CHUNKS = 128;
// block = 4 waves; a wave owns 64 x 8 = 512 consecutive columns of its rows (16-B bf16 / 2 x 16-B fp32 loads); the 4 waves
// of a block take rows r0+w, r0+w+4, ... of the block's row chunk; 4 rows are in flight per wave.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ld, int rows, int cols, float* __restrict__ partial) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  const int rows_per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c + 7 < cols) {
    int r = r0 + w;
    for (; r + 12 < r1; r += 16) {
      float4 v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const T* p = x + (int64_t)(r + 4 * u) * ld + c;
        v[u][0] = ld4<T>(p); v[u][1] = ld4<T>(p + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[0] += v[u][0].x; a[1] += v[u][0].y; a[2] += v[u][0].z; a[3] += v[u][0].w;
        a[4] += v[u][1].x; a[5] += v[u][1].y; a[6] += v[u][1].z; a[7] += v[u][1].w;
      }
    }
    for (; r < r1; r += 4) {
      const T* p = x + (int64_t)r * ld + c;
      const float4 v0 = ld4<T>(p), v1 = ld4<T>(p + 4);
      a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w; a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w;
    }
  } else if (c < cols) {
    for (int r = r0 + w; r < r1; r += 4)
      for (int i = 0; i < 8 && c + i < cols; ++i) a[i] += ldf<T>(x + (int64_t)r * ld + c + i);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[w][lane * 8 + i] = a[i];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int cc = blockIdx.x * 512 + i;
    if (cc < cols) partial[(int64_t)blockIdx.y * cols + cc] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

// ------------------------------------------------------------------ operand preparation
// fp32 Keras kernel W[in][out] -> bf16 copies: wn[in][ldwn] (dgrad operand) and wt[out][ldwt] (forward operand)
// 64 x 64 tiles: 16-B loads, 8-B (4 x bf16) stores in both orientations (128-B row segments per 16 lanes); edge tiles element-wise.
__device__ __forceinline__ void convert_weight_tile(const float* __restrict__ w, int in, int out, bf16_t* __restrict__ wn, int64_t ldwn,
                                                    bf16_t* __restrict__ wt, int64_t ldwt, int i0, int o0, float (*tile)[65]) {
  const int c4 = (threadIdx.x & 15) * 4, r16 = threadIdx.x >> 4;   // 16 lanes x 4 columns, 16 row slots
  const bool full = i0 + 64 <= in && o0 + 64 <= out && (out & 3) == 0 && (ldwn & 3) == 0 && (ldwt & 3) == 0;
  if (full) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r16 + 16 * k;
      const float4 v = *(const float4*)(w + (int64_t)(i0 + r) * out + o0 + c4);
      st4<bf16_t>(wn + (int64_t)(i0 + r) * ldwn + o0 + c4, v);
      tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int o = r16 + 16 * k;   // output row = column o of the tile; this lane writes inputs c4..c4+3 of it
      st4<bf16_t>(wt + (int64_t)(o0 + o) * ldwt + i0 + c4, make_float4(tile[c4][o], tile[c4 + 1][o], tile[c4 + 2][o], tile[c4 + 3][o]));
    }
    return;
  }
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const int i = i0 + r, o = o0 + c;
    float v = 0.f;
    if (i < in && o < out) {
      v = w[(int64_t)i * out + o];
      wn[(int64_t)i * ldwn + o] = (bf16_t)v;
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int o = e >> 6, i = e & 63;
    if (i0 + i < in && o0 + o < out) wt[(int64_t)(o0 + o) * ldwt + i0 + i] = (bf16_t)tile[i][o];
  }
}
__global__ __launch_bounds__(256) void convert_weight_kernel(const float* __restrict__ w, int in, int out, bf16_t* __restrict__ wn,
                                                             int64_t ldwn, bf16_t* __restrict__ wt, int64_t ldwt) {
  __shared__ float tile[64][65];
  convert_weight_tile(w, in, out, wn, ldwn, wt, ldwt, blockIdx.y * 64, blockIdx.x * 64, tile);
}
// Every Dense kernel of a model in ONE launch (the per-step refresh of the bf16 operand copies): 49 launches of 144-576 blocks each left most
// of the chip idle (0.24 ms per ViT-B/16 step at 2.9 TB/s).  blockIdx.x -> (matrix, tile) through the block-offset column of the table.
__global__ __launch_bounds__(256) void convert_weights_batched_kernel(const ConvertDesc* __restrict__ d, int n) {
  __shared__ float tile[64][65];
  __shared__ int sel;
  for (int t = threadIdx.x; t < n; t += 256) {                     // one table entry per thread: a single load latency, not a dependent scan
    const int b0 = d[t].block0, b1 = t + 1 < n ? d[t + 1].block0 : 0x7fffffff;
    if ((int)blockIdx.x >= b0 && (int)blockIdx.x < b1) sel = t;
  }
  __syncthreads();
  const ConvertDesc c = d[sel];
  const int local = (int)blockIdx.x - c.block0, by = local / c.tiles_x, bx = local - by * c.tiles_x;
  convert_weight_tile(c.w, c.in, c.out, c.wn, c.ldwn, c.wt, c.ldwt, by * 64, bx * 64, tile);
}

// out[c][r] = in[r][c]  (bf16, 64x64 tiles through LDS); covers rows x cols exactly (both multiples of 64 by construction)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int64_t ldi, int rows, int cols,
                                                             bf16_t* __restrict__ out, int64_t ldo) {
  __shared__ unsigned short tile[64][66];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const unsigned short* ip = (const unsigned short*)in;
  unsigned short* op = (unsigned short*)out;
  for (int r = ty; r < 64; r += 4) {
    const int rr = r0 + r, cc = c0 + tx;
    tile[r][tx] = (rr < rows && cc < cols) ? ip[(int64_t)rr * ldi + cc] : (unsigned short)0;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int cc = c0 + c, rr = r0 + tx;
    if (cc < cols && rr < rows) op[(int64_t)cc * ldo + rr] = tile[tx][c];
  }
}

template <typename TO>
__global__ void convert_kernel(const float* __restrict__ in, int64_t ldi, TO* __restrict__ out, int64_t ldo, int rows, int cols,
                               int64_t zero_to) {
  const int64_t total = (int64_t)rows * zero_to;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / zero_to;
    const int c = (int)(e - r * zero_to);
    stf<TO>(out + r * ldo + c, c < cols ? in[r * ldi + c] : 0.f);
  }
}

template <typename TI>
__global__ void to_f32_kernel(const TI* __restrict__ in, int64_t ldi, float* __restrict__ out, int64_t ldo, int rows, int cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols;
    const int c = (int)(e - r * cols);
    out[r * ldo + c] = ldf<TI>(in + r * ldi + c);
  }
}

// ------------------------------------------------------------------ pooling (vit.py:170-173)
// x[bi, row, :] = tok[:] for every image (the distillation token appended after the position embedding, distill.py:26-28)
__global__ void set_token_row_kernel(float* __restrict__ x, const float* __restrict__ tok, int b, int ntok, int row, int d) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)b * d) return;
  const int64_t bi = e / d;
  const int c = (int)(e - bi * d);
  x[(bi * ntok + row) * d + c] = tok[c];
}
__global__ void mean_pool_kernel(const float* __restrict__ x, int b, int ntok, int d, float* __restrict__ out, int stride_tok) {
  const int64_t total = (int64_t)b * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bi = e / d;
    const int c = (int)(e - bi * d);
    float a = 0.f;
    for (int t = 0; t < ntok; ++t) a += x[(bi * stride_tok + t) * d + c];
    out[e] = a / (float)ntok;
  }
}
__global__ void mean_pool_bwd_kernel(const float* __restrict__ dp, int b, int ntok, int d, float* __restrict__ g, int stride_tok) {
  const int64_t total = (int64_t)b * ntok * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % d);
    const int64_t bi = e / ((int64_t)ntok * d);
    g[e + bi * (int64_t)(stride_tok - ntok) * d] = dp[bi * d + c] / (float)ntok;
  }
}
// out[j][c] = sum_b g[b][j0+j][c]   (dpos / dcls: vit.py:163-165 VJP)
__global__ void batch_reduce_kernel(const float* __restrict__ g, int b, int ntok, int d, int j0, int nj, float* __restrict__ out) {
  const int64_t total = (int64_t)nj * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e / d;
    const int c = (int)(e - j * d);
    float a = 0.f;
    for (int bi = 0; bi < b; ++bi) a += g[((int64_t)bi * ntok + j0 + j) * d + c];
    out[e] = a;
  }
}
// out[b*np + t][:] = g[b*ntok + tok_off + t][:]  (dE = g[:, 1:, :]) with conversion to T
template <typename TO>
__global__ void extract_rows_kernel(const float* __restrict__ g, int b, int ntok, int tok_off, int np, int d, TO* __restrict__ out,
                                    int64_t ldo) {
  const int64_t total = (int64_t)b * np * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % d);
    const int64_t r = e / d;
    const int64_t bi = r / np;
    const int t = (int)(r - bi * np);
    stf<TO>(out + r * ldo + c, g[((bi * ntok) + tok_off + t) * d + c]);
  }
}
template <typename TO>
__global__ void extract_rows4_kernel(const float* __restrict__ g, int b, int ntok, int tok_off, int np, int d4, TO* __restrict__ out, int64_t ldo) {
  const int64_t total = (int64_t)b * np * d4;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % d4) * 4;
    const int64_t r = e / d4;
    const int64_t bi = r / np;
    const int t = (int)(r - bi * np);
    st4<TO>(out + r * ldo + c, *(const float4*)(g + ((bi * ntok) + tok_off + t) * (int64_t)(d4 * 4) + c));
  }
}
// out[j][c..c+3] = sum_b g[b][j0+j][c..c+3]: one thread per 4 columns, images summed in ascending order (fixed order)
__global__ void batch_reduce4_kernel(const float* __restrict__ g, int b, int ntok, int d4, int j0, int nj, float* __restrict__ out) {
  const int64_t total = (int64_t)nj * d4;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t j = e / d4;
  const int c = (int)(e - j * d4) * 4;
  const int64_t d = (int64_t)d4 * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* p = g + (j0 + j) * d + c;
  const int64_t stride = (int64_t)ntok * d;
  int bi = 0;
  for (; bi + 4 <= b; bi += 4) {   // four independent loads in flight
    const float4 v0 = *(const float4*)(p + (bi + 0) * stride), v1 = *(const float4*)(p + (bi + 1) * stride);
    const float4 v2 = *(const float4*)(p + (bi + 2) * stride), v3 = *(const float4*)(p + (bi + 3) * stride);
    a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
    a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
    a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
    a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
  }
  for (; bi < b; ++bi) { const float4 v = *(const float4*)(p + bi * stride); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  *(float4*)(out + j * d + c) = a;
}
// The same sum with the batch split over the 8 waves of a block (wave w takes images [w*bc, (w+1)*bc), four loads in flight; the partial sums
// meet in LDS and are added in wave order): one thread per output walked b images in 64 dependent round trips (59 us at b = 256 whatever the
// output size; these launches close the backward chain).
__global__ __launch_bounds__(512) void batch_reduce4_split_kernel(const float* __restrict__ g, int b, int ntok, int d4, int j0, int nj,
                                                                  float* __restrict__ out) {
  __shared__ float4 red[8][64];
  const int64_t total = (int64_t)nj * d4;
  const int64_t e = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < total) {
    const int64_t j = e / d4;
    const int c = (int)(e - j * d4) * 4;
    const int64_t d = (int64_t)d4 * 4;
    const float* p = g + (j0 + j) * d + c;
    const int64_t stride = (int64_t)ntok * d;
    const int bc = (b + 7) / 8, b0 = w * bc, b1 = min(b, b0 + bc);
    int bi = b0;
    for (; bi + 4 <= b1; bi += 4) {
      const float4 v0 = *(const float4*)(p + (bi + 0) * stride), v1 = *(const float4*)(p + (bi + 1) * stride);
      const float4 v2 = *(const float4*)(p + (bi + 2) * stride), v3 = *(const float4*)(p + (bi + 3) * stride);
      a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
      a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
      a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
      a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
    }
    for (; bi < b1; ++bi) { const float4 v = *(const float4*)(p + bi * stride); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  }
  red[w][threadIdx.x & 63] = a;
  __syncthreads();
  if (w == 0 && e < total) {
    float4 t = red[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 8; ++k) { const float4 v = red[k][threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    const int64_t j = e / d4;
    *(float4*)(out + j * (int64_t)d4 * 4 + (e - j * d4) * 4) = t;
  }
}
// column sums of a small [rows][d] matrix, rows split over the 4 waves of a block the same way (fixed order)
__global__ __launch_bounds__(256) void sum_rows_split_kernel(const float* __restrict__ in, int rows, int d, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float a = 0.f;
  if (c < d) {
    const int rc = (rows + 3) / 4, r0 = w * rc, r1 = min(rows, r0 + rc);
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
      const float v0 = in[(int64_t)r * d + c], v1 = in[(int64_t)(r + 1) * d + c], v2 = in[(int64_t)(r + 2) * d + c], v3 = in[(int64_t)(r + 3) * d + c];
      a += v0; a += v1; a += v2; a += v3;
    }
    for (; r < r1; ++r) a += in[(int64_t)r * d + c];
  }
  red[w][threadIdx.x & 63] = a;
  __syncthreads();
  if (w == 0 && c < d) out[c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
__global__ void sum_rows_kernel(const float* __restrict__ in, int rows, int d, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  float a = 0.f;
  for (int r = 0; r < rows; ++r) a += in[(int64_t)r * d + c];
  out[c] = a;
}

// ------------------------------------------------------------------ softmax cross-entropy gradient (distill.py:119)
__global__ __launch_bounds__(64) void ce_grad_kernel(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ labels,
                                                     int b, int nc, float inv_batch, float* __restrict__ dlogits,
                                                     float* __restrict__ loss_rows) {
  const int row = blockIdx.x, lane = threadIdx.x;
  if (row >= b) return;
  const float* l = logits + (int64_t)row * ld;
  float m = -INFINITY;
  for (int c = lane; c < nc; c += 64) m = fmaxf(m, l[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < nc; c += 64) s += __expf(l[c] - m);
  s = wave_sum(s);
  const int lab = labels[row];
  const float inv = 1.0f / s;
  for (int c = lane; c < nc; c += 64) dlogits[(int64_t)row * ld + c] = (__expf(l[c] - m) * inv - (c == lab ? 1.f : 0.f)) * inv_batch;
  if (loss_rows && lane == 0) loss_rows[row] = (m + __logf(s) - l[lab]) * inv_batch;
}

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// decoupled-weight-decay Adam (AdamW) / SGD-momentum on the flat fp32 arenas; one fused pass, 16-B accesses
__global__ void adamw_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v, int64_t n4,
                             float lr, float b1, float b2, float eps, float wd, float c1, float c2) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p[e], gg = g[e], mm = m[e], vv = v[e];
    float* P = (float*)&pp; const float* G = (const float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      M[i] = b1 * M[i] + (1.f - b1) * G[i];
      V[i] = b2 * V[i] + (1.f - b2) * G[i] * G[i];
      P[i] = P[i] - lr * ((M[i] * c1) / (sqrtf(V[i] * c2) + eps) + wd * P[i]);
    }
    p[e] = pp; m[e] = mm; v[e] = vv;
  }
}
__global__ void sgd_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, int64_t n4, float lr, float mom, float wd) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p[e], gg = g[e], mm = m ? m[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    float* P = (float*)&pp; const float* G = (const float*)&gg; float* M = (float*)&mm;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = G[i] + wd * P[i];
      M[i] = mom * M[i] + d;
      P[i] -= lr * (m ? M[i] : d);
    }
    p[e] = pp;
    if (m) m[e] = mm;
  }
}
__global__ void fill_zero_kernel(float4* p, int64_t n16) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n16; e += (int64_t)gridDim.x * blockDim.x) p[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void fill_random_bf16_kernel(bf16_t* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = hash32((uint32_t)e * 2654435761U + seed);
    p[e] = (bf16_t)(((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale);  // uniform [-scale, scale)
  }
}
// inverted dropout (vit.py:41,43,64,148; Keras Dropout: y = x*m/(1-rate), m ~ Bernoulli(1-rate)), counter-based
// mask keyed by (seed, site, element) so that backward regenerates it instead of storing it
__device__ __forceinline__ bool dropout_keep(int64_t e, float rate, uint32_t seed_lo, uint32_t seed_hi, uint32_t site) {
  const uint32_t h = hash32(hash32((uint32_t)e ^ seed_lo) + hash32((uint32_t)(e >> 32) ^ seed_hi ^ (site * 0x9e3779b9U)));
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= rate;
}
template <typename T>
__global__ void dropout_kernel(T* x, int64_t n, float rate, uint32_t seed_lo, uint32_t seed_hi, uint32_t site) {
  const float keep_scale = 1.0f / (1.0f - rate);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    stf<T>(x + e, dropout_keep(e, rate, seed_lo, seed_hi, site) ? ldf<T>(x + e) * keep_scale : 0.f);
}
// x_out = resid + f * scale[col]; optionally keeps f (post-dropout) in T for the LayerScale VJP
template <typename T>
__global__ void axpy_resid_kernel(const float* __restrict__ resid, const float* __restrict__ f, const float* __restrict__ scale,
                                  float* __restrict__ out, T* __restrict__ keep, int64_t rows, int d) {
  const int64_t total = rows * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % d);
    const float v = f[e];
    out[e] = resid[e] + v * (scale ? scale[c] : 1.f);
    if (keep) stf<T>(keep + e, v);
  }
}
// out[T] = dropout_mask(g * scale[col])  -- gradient entering a residual branch (LayerScale and/or dropout VJP)
template <typename T>
__global__ void branch_grad_kernel(const float* __restrict__ g, const float* __restrict__ scale, T* __restrict__ out, int64_t rows, int d,
                                   float rate, uint32_t seed_lo, uint32_t seed_hi, uint32_t site) {
  const int64_t total = rows * d;
  const float keep_scale = rate > 0.f ? 1.0f / (1.0f - rate) : 1.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % d);
    float v = g[e] * (scale ? scale[c] : 1.f);
    if (rate > 0.f) v = dropout_keep(e, rate, seed_lo, seed_hi, site) ? v * keep_scale : 0.f;
    stf<T>(out + e, v);
  }
}

// four consecutive columns per lane (d % 4 == 0, fewer than 2^31 elements, 16-B friendly pointers): one 16-B read, one 8/16-B store and
// one 32-bit modulo per four elements instead of a 64-bit modulo and a 2-byte store per element (CaiT: 52 launches per step at 2 TB/s)
template <typename T>
__global__ void branch_grad4_kernel(const float4* __restrict__ g, const float4* __restrict__ scale, T* __restrict__ out, uint32_t total4, uint32_t d4,
                                    float rate, uint32_t seed_lo, uint32_t seed_hi, uint32_t site) {
  const float keep_scale = rate > 0.f ? 1.0f / (1.0f - rate) : 1.f;
  for (uint32_t e4 = blockIdx.x * blockDim.x + threadIdx.x; e4 < total4; e4 += gridDim.x * blockDim.x) {
    float4 v = g[e4];
    if (scale) { const float4 sc = scale[e4 % d4]; v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
    if (rate > 0.f) {
      const int64_t e = (int64_t)e4 * 4;
      v.x = dropout_keep(e, rate, seed_lo, seed_hi, site) ? v.x * keep_scale : 0.f;
      v.y = dropout_keep(e + 1, rate, seed_lo, seed_hi, site) ? v.y * keep_scale : 0.f;
      v.z = dropout_keep(e + 2, rate, seed_lo, seed_hi, site) ? v.z * keep_scale : 0.f;
      v.w = dropout_keep(e + 3, rate, seed_lo, seed_hi, site) ? v.w * keep_scale : 0.f;
    }
    st4<T>(out + (int64_t)e4 * 4, v);
  }
}

template <typename T>
__global__ void resid_add_kernel(const float* __restrict__ resid, const T* __restrict__ t, float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = resid[e] + ldf<T>(t + e);
}

inline int grid_for(int64_t total, int block = 256) { return (int)std::min<int64_t>(ceil_div(total, block), 256 * 8); }

}  // namespace

void launch_unfold(const float* img, void* out, int out_bf16, int b, int H, int W, int C, int ph, int pw, int64_t ldo, hipStream_t s) {
  const int64_t total = (int64_t)b * (H / ph) * (W / pw) * ldo;
  if (total == 0) return;
  const bool vec = ((pw * C) % 4 == 0) && (ldo % 4 == 0) && ((W * C) % 4 == 0) && (((uintptr_t)img) % 16 == 0) && (((uintptr_t)out) % 16 == 0) &&
                   ldo < (1 << 30);
  if (vec) {
    if (out_bf16) hipLaunchKernelGGL(unfold4_kernel<bf16_t>, dim3(grid_for(total / 4)), dim3(256), 0, s, img, (bf16_t*)out, b, H, W, C, ph, pw, (int)ldo);
    else hipLaunchKernelGGL(unfold4_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, img, (float*)out, b, H, W, C, ph, pw, (int)ldo);
    return;
  }
  if (out_bf16) hipLaunchKernelGGL(unfold_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, img, (bf16_t*)out, b, H, W, C, ph, pw, ldo);
  else hipLaunchKernelGGL(unfold_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, img, (float*)out, b, H, W, C, ph, pw, ldo);
}
void launch_fold_add(const float* dp, int64_t ld, float* dimg, int b, int H, int W, int C, int ph, int pw, hipStream_t s) {
  const int64_t total = (int64_t)b * H * W * C;
  if (total == 0) return;
  hipLaunchKernelGGL(fold_kernel, dim3(grid_for(total)), dim3(256), 0, s, dp, ld, dimg, b, H, W, C, ph, pw);
}
void launch_cls_pos_row(float* x, const float* cls, const float* pos, int b, int ntok, int d, int64_t ldx, hipStream_t s) {
  hipLaunchKernelGGL(cls_pos_row_kernel, dim3(grid_for((int64_t)b * d)), dim3(256), 0, s, x, cls, pos, b, ntok, d, ldx);
}

#define VITX_VPL_DISPATCH(d, CALL)                      \
  do {                                                  \
    const int vpl_ = (int)ceil_div((d), 256);           \
    if (vpl_ <= 1) { CALL(1); }                         \
    else if (vpl_ == 2) { CALL(2); }                    \
    else if (vpl_ == 3) { CALL(3); }                    \
    else if (vpl_ == 4) { CALL(4); }                    \
    else if (vpl_ <= 6) { CALL(6); }                    \
    else if (vpl_ <= 8) { CALL(8); }                    \
    else { CALL(16); }                                  \
  } while (0)

// the four-columns-per-lane LayerNorm kernels need d % 4 == 0 and 16-B aligned rows everywhere; anything else takes the any-width forms
static bool ln_vec_ok(int d, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
  if (d % 4) return false;
  for (int64_t l : lds) if (l % 4) return false;
  for (const void* p : ptrs) if (((uintptr_t)p) % 16) return false;
  return true;
}

void launch_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int y_bf16, int64_t ldy,
                          float* mean, float* rstd, int rows, int d, float eps, hipStream_t s) {
  if (rows == 0) return;
  dim3 grid((unsigned)ceil_div(rows, 4)), block(256);
  if (!ln_vec_ok(d, {ldx, ldy}, {x, gamma, beta, y})) {
    if (y_bf16) hipLaunchKernelGGL(layernorm_fwd_any_kernel<bf16_t>, grid, block, 0, s, x, ldx, gamma, beta, (bf16_t*)y, ldy, mean, rstd, rows, d, eps);
    else hipLaunchKernelGGL(layernorm_fwd_any_kernel<float>, grid, block, 0, s, x, ldx, gamma, beta, (float*)y, ldy, mean, rstd, rows, d, eps);
    return;
  }
#define CALL(V)                                                                                                             \
  if (y_bf16) hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, V>), grid, block, 0, s, x, ldx, gamma, beta, (bf16_t*)y, ldy, \
                                 mean, rstd, rows, d, eps);                                                                 \
  else hipLaunchKernelGGL((layernorm_fwd_kernel<float, V>), grid, block, 0, s, x, ldx, gamma, beta, (float*)y, ldy, mean, rstd, rows, d, eps)
  VITX_VPL_DISPATCH(d, CALL);
#undef CALL
}

int64_t layernorm_bwd_ws_elems(int d) { return (int64_t)(LNB_BLOCKS_MAX + 32) * 4 * d; }   // (4 partial rows per block: the LayerScale-fused form)

// The parameter-gradient sums (dgamma, dbeta, the optional column sums of g_in) are a two-level reduction of per-block partials that nothing in
// the backward chain consumes: with `deferred` != nullptr the main kernel(s) only are launched and *deferred receives the number of partial rows;
// the caller then runs launch_layernorm_bwd_reduce on a stream of its choice (the engine: the side stream, beside the weight gradients).
void launch_layernorm_bwd(const void* dy, int dy_bf16, int64_t lddy, const float* x, int64_t ldx, const float* mean, const float* rstd,
                          const float* gamma, const float* g_in, int64_t ldgi, float* g_out, int64_t ldgo, void* g_lp, int64_t ldglp,
                          float* partial_ws, float* dgamma, float* dbeta, float* gsum, int rows, int d, hipStream_t s, int* deferred) {
  if (deferred) *deferred = 0;
  if (rows == 0) return;
  const int want_gsum = (gsum != nullptr && g_in != nullptr) ? 1 : 0;
  if (!ln_vec_ok(d, {lddy, ldx, g_in ? ldgi : 0, ldgo, g_lp ? ldglp : 0}, {dy, x, gamma, g_in, g_out, g_lp})) {
    dim3 rgrid((unsigned)ceil_div(rows, 4)), block(256);
    const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(LNB_BLOCKS, ceil_div(rows, 64)));
    dim3 cgrid((unsigned)ceil_div(d, 256), (unsigned)chunks);
    if (dy_bf16) {
      hipLaunchKernelGGL((layernorm_bwd_any_cols_kernel<bf16_t>), cgrid, block, 0, s, (const bf16_t*)dy, lddy, x, ldx, mean, rstd, g_in, ldgi, partial_ws, rows, d, want_gsum);
      hipLaunchKernelGGL((layernorm_bwd_any_rows_kernel<bf16_t, bf16_t>), rgrid, block, 0, s, (const bf16_t*)dy, lddy, x, ldx, mean, rstd, gamma, g_in, ldgi, g_out, ldgo,
                         (bf16_t*)g_lp, ldglp, rows, d);
    } else {
      hipLaunchKernelGGL((layernorm_bwd_any_cols_kernel<float>), cgrid, block, 0, s, (const float*)dy, lddy, x, ldx, mean, rstd, g_in, ldgi, partial_ws, rows, d, want_gsum);
      hipLaunchKernelGGL((layernorm_bwd_any_rows_kernel<float, float>), rgrid, block, 0, s, (const float*)dy, lddy, x, ldx, mean, rstd, gamma, g_in, ldgi, g_out, ldgo,
                         (float*)g_lp, ldglp, rows, d);
    }
    // (the column kernel runs FIRST: g_out may alias g_in, which it reads)
    if (deferred) *deferred = chunks;
    else launch_layernorm_bwd_reduce(partial_ws, chunks, d, dgamma, dbeta, want_gsum ? gsum : nullptr, s);
    return;
  }
  // VITX_LN_CORESIDENT=1 (experiment, off by default): bf16 rows of 768 floats take the 4-wave, 96-register form that fits beside a resident
  // weight-gradient GEMM.  Measured (profiles/r6/coresidency_r6.md): with 1024 blocks it is as fast as the 8-wave form and the step is unchanged
  // (its blocks are placed first and fill the register file themselves); with ONE block per CU (VITX_LN_CO_BLOCKS=256), the only geometry that
  // leaves the GEMM workgroup its registers, the weight gradients do progress underneath (their kernel intervals shrink by a quarter) but four
  // waves per CU pull a third of the bandwidth: 338 us per pass instead of 101, step +0.7 ms.
  static const int co_env = [] { const char* v = vitx_env("VITX_LN_CORESIDENT"); return v ? atoi(v) : 0; }();
  if (co_env && dy_bf16 && ceil_div(d, 256) == 3) {
    static const int co_blocks = [] { const char* v = vitx_env("VITX_LN_CO_BLOCKS"); return v ? std::max(1, std::min(LNB_BLOCKS_MAX, atoi(v))) : LNB_BLOCKS_MAX; }();
    const int nb = (int)std::min<int64_t>(co_blocks, ceil_div(rows, 4));
    hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, bf16_t, 3, false, true>), dim3(nb), dim3(256), (size_t)4 * d * sizeof(float), s, (const bf16_t*)dy, lddy, x, ldx,
                       mean, rstd, gamma, g_in, ldgi, g_out, ldgo, (bf16_t*)g_lp, ldglp, partial_ws, rows, d, want_gsum, (const bf16_t*)nullptr, (int64_t)0,
                       (const float*)nullptr);
    if (deferred) *deferred = nb;
    else launch_layernorm_bwd_reduce(partial_ws, nb, d, dgamma, dbeta, want_gsum ? gsum : nullptr, s);
    return;
  }
  const int nblk = (int)std::min<int64_t>(LNB_BLOCKS, ceil_div(rows, 8));
  dim3 grid(nblk), block(LNB_THREADS);
  const size_t shm = (size_t)(LNB_THREADS / 64) * d * sizeof(float);
#define CALL(V)                                                                                                                  \
  if (dy_bf16) hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, bf16_t, V, false>), grid, block, shm, s, (const bf16_t*)dy, lddy, x, ldx, \
                                  mean, rstd, gamma, g_in, ldgi, g_out, ldgo, (bf16_t*)g_lp, ldglp, partial_ws, rows, d, want_gsum, \
                                  (const bf16_t*)nullptr, (int64_t)0, (const float*)nullptr); \
  else hipLaunchKernelGGL((layernorm_bwd_kernel<float, float, V, false>), grid, block, shm, s, (const float*)dy, lddy, x, ldx, mean, rstd, \
                          gamma, g_in, ldgi, g_out, ldgo, (float*)g_lp, ldglp, partial_ws, rows, d, want_gsum, (const float*)nullptr, (int64_t)0, \
                          (const float*)nullptr)
  VITX_VPL_DISPATCH(d, CALL);
#undef CALL
  // partial layout [blk][3][d]: dgamma = sum_blk partial[blk][0], dbeta = sum_blk partial[blk][1]
  if (deferred) *deferred = nblk;
  else launch_layernorm_bwd_reduce(partial_ws, nblk, d, dgamma, dbeta, want_gsum ? gsum : nullptr, s);
}
// ---- LayerNorm VJP + the LayerScale VJP of the branch that consumes its result (bf16 mode; see layernorm_bwd_kernel, FUSE)
bool layernorm_bwd_scale_ok(int d) { return d % 4 == 0 && d <= 8 * 256; }
void launch_layernorm_bwd_scale(const bf16_t* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                                const float* g_in, int64_t ldgi, float* g_out, int64_t ldgo, bf16_t* dbr, int64_t lddbr, const bf16_t* nf, int64_t ldnf,
                                const float* nscale, float* partial_ws, int rows, int d, hipStream_t s, int* nparts) {
  *nparts = 0;
  if (rows == 0) return;
  const int nblk = (int)std::min<int64_t>(LNB_BLOCKS, ceil_div(rows, 8));
  dim3 grid(nblk), block(LNB_THREADS);
  const size_t shm = (size_t)(LNB_THREADS / 64) * d * sizeof(float);
#define CALL(V)                                                                                                                              \
  hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, bf16_t, V, true>), grid, block, shm, s, dy, lddy, x, ldx, mean, rstd, gamma, g_in, ldgi, g_out, \
                     ldgo, dbr, lddbr, partial_ws, rows, d, 0, nf, ldnf, nscale)
  VITX_VPL_DISPATCH(d, CALL);
#undef CALL
  *nparts = nblk;
}
__global__ void mul_vec_kernel(float* __restrict__ a, const float* __restrict__ b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] *= b[i];
}
// partial layout [blk][4][d]: dgamma, dbeta, column sums of g, column sums of g * f; dbias (optional) = nscale * colsum(g)
void launch_layernorm_bwd_scale_reduce(float* partial_ws, int nparts, int d, float* dgamma, float* dbeta, float* dscale, float* dbias, const float* nscale,
                                       hipStream_t s) {
  if (nparts <= 0) return;
  float* ws2 = partial_ws + (int64_t)LNB_BLOCKS_MAX * 4 * d;
  launch_reduce_partials3(partial_ws, nparts, (int64_t)4 * d, d, 2, dgamma, dbeta, nullptr, ws2, 1.0f, s);
  if (dbias) {
    launch_reduce_partials3(partial_ws + 2 * (int64_t)d, nparts, (int64_t)4 * d, d, 2, dbias, dscale, nullptr, ws2, 1.0f, s);
    hipLaunchKernelGGL(mul_vec_kernel, dim3((unsigned)ceil_div(d, 256)), dim3(256), 0, s, dbias, nscale, d);
  } else {
    launch_reduce_partials3(partial_ws + 3 * (int64_t)d, nparts, (int64_t)4 * d, d, 1, dscale, nullptr, nullptr, ws2, 1.0f, s);
  }
}

void launch_layernorm_bwd_reduce(float* partial_ws, int nparts, int d, float* dgamma, float* dbeta, float* gsum, hipStream_t s) {
  if (nparts <= 0) return;
  launch_reduce_partials3(partial_ws, nparts, (int64_t)3 * d, d, gsum ? 3 : 2, dgamma, dbeta, gsum, partial_ws + (int64_t)LNB_BLOCKS_MAX * 3 * d, 1.0f, s);
}

// level-2 scratch for the two-level path lives behind the level-1 partials (callers size their workspace with *_ws_elems)
void launch_reduce_partials3(const float* partial, int nparts, int64_t stride, int64_t seg, int nseg, float* out0, float* out1, float* out2,
                             float* ws2, float alpha, hipStream_t s) {
  const int64_t n = seg * nseg;
  if (n == 0) return;
  const unsigned gx = (unsigned)ceil_div(n, 64);
  if (nparts > 32 && ws2 != nullptr) {
    const int chunks = (int)std::min<int64_t>(32, ceil_div(nparts, 8));
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, chunks), dim3(256), 0, s, partial, nparts, stride, n, ws2, nullptr, nullptr, seg, n, 1.0f);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, s, ws2, chunks, n, n, out0, out1, out2, seg, (int64_t)0, alpha);
  } else {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, s, partial, nparts, stride, n, out0, out1, out2, seg, (int64_t)0, alpha);
  }
}
void launch_reduce_partials(const float* partial, int nparts, int64_t stride, int64_t n, float* out, float alpha, hipStream_t s) {
  if (n >= 4096 && nparts <= 32 && (n & 3) == 0 && (stride & 3) == 0 && ((uintptr_t)partial) % 16 == 0 && ((uintptr_t)out) % 16 == 0) {
    hipLaunchKernelGGL(reduce_partials4_kernel, dim3((unsigned)ceil_div(n / 4, 256)), dim3(256), 0, s, (const float4*)partial, nparts, stride / 4,
                       n / 4, (float4*)out, alpha);
    return;
  }
  launch_reduce_partials3(partial, nparts, stride, n, 1, out, nullptr, nullptr, nullptr, alpha, s);
}

int64_t colsum_ws_elems(int cols) { return (int64_t)(CS_CHUNKS + 32) * cols; }
void launch_colsum(const void* x, int is_bf16, int64_t ld, int rows, int cols, float* partial_ws, float* out, hipStream_t s) {
  // ~2048 blocks when the matrix is large; ld and the base pointer are 16-B aligned for every internal buffer
  const int cblocks = (int)ceil_div(cols, 512);
  const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)CS_CHUNKS, ceil_div(rows, 16), ceil_div(2048, cblocks)}));
  dim3 grid((unsigned)cblocks, chunks), block(256);
  if (is_bf16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, ld, rows, cols, partial_ws);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)x, ld, rows, cols, partial_ws);
  launch_reduce_partials3(partial_ws, chunks, cols, cols, 1, out, nullptr, nullptr, partial_ws + (int64_t)CS_CHUNKS * cols, 1.0f, s);
}

void launch_convert_weight(const float* w, int in, int out, bf16_t* wn, int64_t ldwn, bf16_t* wt, int64_t ldwt, hipStream_t s) {
  dim3 grid((unsigned)ceil_div(out, 64), (unsigned)ceil_div(in, 64)), block(256);
  hipLaunchKernelGGL(convert_weight_kernel, grid, block, 0, s, w, in, out, wn, ldwn, wt, ldwt);
}
void launch_convert_weights_batched(const ConvertDesc* descs_dev, int n, int total_blocks, hipStream_t s) {
  if (n > 0 && total_blocks > 0) hipLaunchKernelGGL(convert_weights_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, s, descs_dev, n);
}
void launch_transpose_bf16(const bf16_t* in, int64_t ldi, int rows, int cols, bf16_t* out, int64_t ldo, hipStream_t s) {
  dim3 grid((unsigned)ceil_div(cols, 64), (unsigned)ceil_div(rows, 64)), block(256);
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, block, 0, s, in, ldi, rows, cols, out, ldo);
}
void launch_convert(const float* in, int64_t ldi, void* out, int out_bf16, int64_t ldo, int rows, int cols, int64_t zero_to,
                    hipStream_t s) {
  const int64_t total = (int64_t)rows * zero_to;
  if (total == 0) return;
  if (out_bf16) hipLaunchKernelGGL(convert_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, in, ldi, (bf16_t*)out, ldo, rows, cols, zero_to);
  else hipLaunchKernelGGL(convert_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, in, ldi, (float*)out, ldo, rows, cols, zero_to);
}
void launch_to_f32(const void* in, int in_bf16, int64_t ldi, float* out, int64_t ldo, int rows, int cols, hipStream_t s) {
  const int64_t total = (int64_t)rows * cols;
  if (total == 0) return;
  if (in_bf16) hipLaunchKernelGGL(to_f32_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)in, ldi, out, ldo, rows, cols);
  else hipLaunchKernelGGL(to_f32_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)in, ldi, out, ldo, rows, cols);
}
void launch_set_token_row(float* x, const float* tok, int b, int ntok, int row, int d, hipStream_t s) {
  hipLaunchKernelGGL(set_token_row_kernel, dim3((unsigned)ceil_div((int64_t)b * d, 256)), dim3(256), 0, s, x, tok, b, ntok, row, d);
}
void launch_mean_pool(const float* x, int b, int ntok, int d, float* out, hipStream_t s, int stride_tok) {
  hipLaunchKernelGGL(mean_pool_kernel, dim3(grid_for((int64_t)b * d)), dim3(256), 0, s, x, b, ntok, d, out, stride_tok ? stride_tok : ntok);
}
void launch_mean_pool_bwd(const float* dp, int b, int ntok, int d, float* g, hipStream_t s, int stride_tok) {
  hipLaunchKernelGGL(mean_pool_bwd_kernel, dim3(grid_for((int64_t)b * ntok * d)), dim3(256), 0, s, dp, b, ntok, d, g, stride_tok ? stride_tok : ntok);
}
void launch_batch_reduce(const float* g, int b, int ntok, int d, int j0, int nj, float* out, hipStream_t s) {
  if (nj <= 0) return;
  if (d % 4 == 0 && ((uintptr_t)g) % 16 == 0 && ((uintptr_t)out) % 16 == 0) {
    if (b >= 32) hipLaunchKernelGGL(batch_reduce4_split_kernel, dim3((unsigned)ceil_div((int64_t)nj * (d / 4), 64)), dim3(512), 0, s, g, b, ntok, d / 4, j0, nj, out);
    else hipLaunchKernelGGL(batch_reduce4_kernel, dim3((unsigned)ceil_div((int64_t)nj * (d / 4), 64)), dim3(64), 0, s, g, b, ntok, d / 4, j0, nj, out);
    return;
  }
  hipLaunchKernelGGL(batch_reduce_kernel, dim3(grid_for((int64_t)nj * d)), dim3(256), 0, s, g, b, ntok, d, j0, nj, out);
}
void launch_extract_rows(const float* g, int b, int ntok, int tok_off, int np, int d, void* out, int out_bf16, int64_t ldo, hipStream_t s) {
  const int64_t total = (int64_t)b * np * d;
  if (total == 0) return;
  if (d % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)g) % 16 == 0 && ((uintptr_t)out) % 16 == 0) {
    if (out_bf16) hipLaunchKernelGGL(extract_rows4_kernel<bf16_t>, dim3(grid_for(total / 4)), dim3(256), 0, s, g, b, ntok, tok_off, np, d / 4, (bf16_t*)out, ldo);
    else hipLaunchKernelGGL(extract_rows4_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, s, g, b, ntok, tok_off, np, d / 4, (float*)out, ldo);
    return;
  }
  if (out_bf16) hipLaunchKernelGGL(extract_rows_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, g, b, ntok, tok_off, np, d, (bf16_t*)out, ldo);
  else hipLaunchKernelGGL(extract_rows_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, g, b, ntok, tok_off, np, d, (float*)out, ldo);
}
void launch_sum_rows(const float* in, int rows, int d, float* out, hipStream_t s) {
  if (rows >= 32) hipLaunchKernelGGL(sum_rows_split_kernel, dim3((unsigned)ceil_div(d, 64)), dim3(256), 0, s, in, rows, d, out);
  else hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)ceil_div(d, 256)), dim3(256), 0, s, in, rows, d, out);
}
void launch_ce_grad(const float* logits, int64_t ld, const int32_t* labels, int b, int nc, float inv_batch, float* dlogits, float* loss,
                    hipStream_t s) {
  hipLaunchKernelGGL(ce_grad_kernel, dim3(b), dim3(64), 0, s, logits, ld, labels, b, nc, inv_batch, dlogits, loss);
}
void launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, int step, hipStream_t s) {
  if (n == 0) return;
  const float c1 = 1.0f / (1.0f - powf(b1, (float)step)), c2 = 1.0f / (1.0f - powf(b2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, (float4*)p, (const float4*)g, (float4*)m, (float4*)v, n / 4, lr, b1, b2, eps, wd, c1, c2);
}
void launch_sgd(float* p, const float* g, float* m, int64_t n, float lr, float mom, float wd, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, (float4*)p, (const float4*)g, (float4*)m, n / 4, lr, mom, wd);
}
void launch_fill_zero(void* p, int64_t bytes, hipStream_t s) {
  if (bytes <= 0) return;
  hipLaunchKernelGGL(fill_zero_kernel, dim3(grid_for(bytes / 16)), dim3(256), 0, s, (float4*)p, bytes / 16);
}
// max over rows < `rows`, cols < `cols` of |a - b| / (1 + |b|), as the bit pattern of a non-negative float (atomicMax on uint32 keeps the order)
template <typename T>
__global__ __launch_bounds__(256) void max_rel_diff_kernel(const T* a, const T* b, int64_t rows, int cols, int64_t lda, int64_t ldb, unsigned* out) {
  float m = 0.f;
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const float x = ldf<T>(a + r * lda + c), y = ldf<T>(b + r * ldb + c);
    const float d = fabsf(x - y) / (1.f + fabsf(y));
    m = fmaxf(m, (d == d) ? d : INFINITY);   // NaN anywhere -> inf
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned, m));
}
void launch_max_rel_diff(const void* a, const void* b, int is_bf16, int64_t rows, int cols, int64_t lda, int64_t ldb, float* out_max, hipStream_t s) {
  const unsigned grid = (unsigned)std::min<int64_t>(4096, std::max<int64_t>(1, ceil_div(rows * cols, 256)));
  if (is_bf16) hipLaunchKernelGGL(max_rel_diff_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, rows, cols, lda, ldb, (unsigned*)out_max);
  else hipLaunchKernelGGL(max_rel_diff_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, rows, cols, lda, ldb, (unsigned*)out_max);
}
void launch_fill_random_bf16(bf16_t* p, int64_t n, uint32_t seed, float scale, hipStream_t s) {
  hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, n, seed, scale);
}
void launch_dropout(void* x, int is_bf16, int64_t n, float rate, uint64_t seed, uint32_t site, hipStream_t s) {
  if (n == 0 || rate <= 0.f) return;
  if (is_bf16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, (bf16_t*)x, n, rate, (uint32_t)seed, (uint32_t)(seed >> 32), site);
  else hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (float*)x, n, rate, (uint32_t)seed, (uint32_t)(seed >> 32), site);
}
void launch_axpy_resid(const float* resid, const float* f, const float* scale, float* out, void* keep, int keep_bf16, int64_t rows, int d,
                       hipStream_t s) {
  if (rows == 0) return;
  if (keep_bf16) hipLaunchKernelGGL(axpy_resid_kernel<bf16_t>, dim3(grid_for(rows * d)), dim3(256), 0, s, resid, f, scale, out, (bf16_t*)keep, rows, d);
  else hipLaunchKernelGGL(axpy_resid_kernel<float>, dim3(grid_for(rows * d)), dim3(256), 0, s, resid, f, scale, out, (float*)keep, rows, d);
}
void launch_branch_grad(const float* g, const float* scale, void* out, int out_bf16, int64_t rows, int d, float rate, uint64_t seed,
                        uint32_t site, hipStream_t s) {
  if (rows == 0) return;
  if (d % 4 == 0 && rows * d < (1ll << 31) && ((uintptr_t)g) % 16 == 0 && ((uintptr_t)out) % 16 == 0 && ((uintptr_t)scale) % 16 == 0) {
    const uint32_t total4 = (uint32_t)(rows * d / 4);
    if (out_bf16) hipLaunchKernelGGL(branch_grad4_kernel<bf16_t>, dim3(grid_for(total4)), dim3(256), 0, s, (const float4*)g, (const float4*)scale, (bf16_t*)out, total4, (uint32_t)(d / 4), rate, (uint32_t)seed, (uint32_t)(seed >> 32), site);
    else hipLaunchKernelGGL(branch_grad4_kernel<float>, dim3(grid_for(total4)), dim3(256), 0, s, (const float4*)g, (const float4*)scale, (float*)out, total4, (uint32_t)(d / 4), rate, (uint32_t)seed, (uint32_t)(seed >> 32), site);
    return;
  }
  if (out_bf16) hipLaunchKernelGGL(branch_grad_kernel<bf16_t>, dim3(grid_for(rows * d)), dim3(256), 0, s, g, scale, (bf16_t*)out, rows, d, rate, (uint32_t)seed, (uint32_t)(seed >> 32), site);
  else hipLaunchKernelGGL(branch_grad_kernel<float>, dim3(grid_for(rows * d)), dim3(256), 0, s, g, scale, (float*)out, rows, d, rate, (uint32_t)seed, (uint32_t)(seed >> 32), site);
}
void launch_resid_add(const float* resid, const void* t, int t_bf16, float* out, int64_t n, hipStream_t s) {
  if (n == 0) return;
  if (t_bf16) hipLaunchKernelGGL(resid_add_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, resid, (const bf16_t*)t, out, n);
  else hipLaunchKernelGGL(resid_add_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, resid, (const float*)t, out, n);
}

// ------------------------------------------------------------------------------------------------
// T2T tokenizer (t2t.py:39-47): tf.image.extract_patches(x, sizes=[1,k,k,1], strides=[1,s,s,1], rates=[1,1,1,1], padding='SAME')
// on NHWC x.  Output [b, oh, ow, k*k*C] with oh = ceil(H/s), the k x k window of output pixel (oi, oj) starts at
// (oi*s - pad_top, oj*s - pad_left), pad_total = max((o-1)*s + k - in, 0), pad_before = pad_total / 2 (TensorFlow's SAME rule: the
// odd pixel goes to the bottom / right), out-of-image taps read as 0.  Feature order (ki, kj, c).  Pure index arithmetic: bit-exact.
// One thread per output element: consecutive threads walk (ki, kj, c), i.e. k*C-float contiguous runs of an input row.
// ------------------------------------------------------------------------------------------------
namespace {
// division by a run-time constant as multiply-high + add + shift (exact for n < 2^31): the index chains below would otherwise spend
// ~400 instructions per 4-byte element on 64-bit divides (first version: 1.3-1.5 TB/s, ALU-bound)
struct FastDiv {
  uint32_t d, m, s;
  FastDiv() : d(1), m(1), s(0) {}
  explicit FastDiv(uint32_t dd) : d(dd) {
    s = 0;
    while ((1ull << s) < dd) ++s;
    m = (uint32_t)(((1ull << 32) * ((1ull << s) - dd)) / dd + 1);
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (__umulhi(n, m) + n) >> s; }
};

// elements [0, n) of the output rows [row0, ...): n < 2^31 per launch (the host splits larger tensors at row boundaries)
template <bool VEC>
__global__ void __launch_bounds__(256) extract_patches_kernel(const float* __restrict__ x, float* __restrict__ out, uint32_t n, uint32_t row0,
                                                              FastDiv feat, FastDiv run, FastDiv ow, FastDiv oh, int H, int W, int C, int k, int st,
                                                              int pt, int pl) {
  const uint32_t stride = gridDim.x * blockDim.x;
  auto src_of = [&](uint32_t e, bool& ok) -> int64_t {
    const uint32_t rl = feat.div(e), f = e - rl * feat.d;
    const uint32_t ki = run.div(f), r = f - ki * run.d;             // r = kj*C + c
    const uint32_t row = row0 + rl;
    const uint32_t t = ow.div(row), oj = row - t * ow.d;
    const uint32_t bi = oh.div(t), oi = t - bi * oh.d;
    const int y = (int)oi * st - pt + (int)ki, x0 = (int)oj * st - pl;
    const int lo = max(0, -x0) * C, hi = min(k, W - x0) * C;          // taps kj with 0 <= x0 + kj < W, as a range of r
    ok = y >= 0 && y < H && (int)r >= lo && (int)r < hi;
    return (((int64_t)bi * H + y) * W + x0) * C + (int)r;
  };
  if (VEC) {   // four consecutive output floats per lane, one 16-B store (the output is 2-3x the input: stores dominate)
    const uint32_t n4 = n >> 2;
    for (uint32_t e4 = blockIdx.x * blockDim.x + threadIdx.x; e4 < n4; e4 += stride) {
      bool k0, k1, k2, k3;
      const uint32_t e = e4 << 2;
      const int64_t s0 = src_of(e, k0), s1 = src_of(e + 1, k1), s2 = src_of(e + 2, k2), s3 = src_of(e + 3, k3);
      *(float4*)(out + e) = make_float4(k0 ? x[s0] : 0.f, k1 ? x[s1] : 0.f, k2 ? x[s2] : 0.f, k3 ? x[s3] : 0.f);
    }
    const uint32_t e = (n4 << 2) + blockIdx.x * blockDim.x + threadIdx.x;     // the last n % 4 elements
    if (e < n) {
      bool ok;
      const int64_t s0 = src_of(e, ok);
      out[e] = ok ? x[s0] : 0.f;
    }
    return;
  }
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  for (; e < n && (uint64_t)e + 3ull * stride < n; e += 4 * stride) {   // four independent loads in flight per lane
    bool k0, k1, k2, k3;
    const int64_t s0 = src_of(e, k0), s1 = src_of(e + stride, k1), s2 = src_of(e + 2 * stride, k2), s3 = src_of(e + 3 * stride, k3);
    const float v0 = k0 ? x[s0] : 0.f, v1 = k1 ? x[s1] : 0.f, v2 = k2 ? x[s2] : 0.f, v3 = k3 ? x[s3] : 0.f;
    out[e] = v0; out[e + stride] = v1; out[e + 2 * stride] = v2; out[e + 3 * stride] = v3;
  }
  for (; e < n; e += stride) {
    bool ok;
    const int64_t s0 = src_of(e, ok);
    out[e] = ok ? x[s0] : 0.f;
  }
}

// VJP: dx[b, y, xx, c] = sum over the windows that cover (y, xx): oi with 0 <= y + pt - oi*st < k, likewise oj, of
// dout[b, oi, oj, (ki, kj, c)] -- a gather per input element, windows visited in ascending (ki, kj): fixed summation order
__global__ void __launch_bounds__(256) extract_patches_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, uint32_t n, uint32_t pix0,
                                                                  FastDiv Cd, FastDiv Wd, FastDiv Hd, FastDiv sd, int k, int st, int oh, int ow,
                                                                  int pt, int pl) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const int C = (int)Cd.d, feat = k * k * C;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const uint32_t pl_ = Cd.div(e), c = e - pl_ * Cd.d;
    const uint32_t pix = pix0 + pl_;
    const uint32_t t = Wd.div(pix), xx = pix - t * Wd.d;
    const uint32_t bi = Hd.div(t), y = t - bi * Hd.d;
    const int ay = (int)y + pt, ax = (int)xx + pl;
    const int oi_hi = min((int)sd.div((uint32_t)ay), oh - 1), oj_hi = min((int)sd.div((uint32_t)ax), ow - 1);
    const int oi_lo = ay - k + 1 > 0 ? (int)sd.div((uint32_t)(ay - k + st)) : 0;      // ceil((ay - k + 1) / st)
    const int oj_lo = ax - k + 1 > 0 ? (int)sd.div((uint32_t)(ax - k + st)) : 0;
    float a = 0.f;
    for (int oi = oi_hi; oi >= oi_lo; --oi) {          // descending window index = ascending tap index
      const int ki = ay - oi * st;
      const float* rowp = dout + (((int64_t)bi * oh + oi) * ow) * (int64_t)feat + (int64_t)ki * k * C + c;
      for (int oj = oj_hi; oj >= oj_lo; --oj) a += rowp[(int64_t)oj * feat + (ax - oj * st) * C];
    }
    dx[(uint64_t)pl_ * Cd.d + c] = a;
  }
}
inline int grid_for_ep(int64_t total) { return (int)std::min<int64_t>(ceil_div(total, 256), 256 * 8); }
}  // namespace

void extract_patches_geometry(int H, int W, int k, int st, int* oh, int* ow, int* pt, int* pl) {
  *oh = (H + st - 1) / st;
  *ow = (W + st - 1) / st;
  const int ph = std::max((*oh - 1) * st + k - H, 0), pw = std::max((*ow - 1) * st + k - W, 0);
  *pt = ph / 2;
  *pl = pw / 2;
}
void launch_extract_patches(const float* x, float* out, int b, int H, int W, int C, int k, int st, hipStream_t s) {
  int oh, ow, pt, pl;
  extract_patches_geometry(H, W, k, st, &oh, &ow, &pt, &pl);
  const int64_t feat = (int64_t)k * k * C, rows = (int64_t)b * oh * ow;
  if (rows == 0) return;
  // 32-bit element indices inside a launch: at most 2^30 elements (whole rows) per launch
  const int64_t rows_per = std::max<int64_t>(1, (1ll << 30) / feat);
  for (int64_t r0 = 0; r0 < rows; r0 += rows_per) {
    const int64_t nr = std::min(rows_per, rows - r0), n = nr * feat;
    float* o = out + r0 * feat;
    if (((uintptr_t)o) % 16 == 0 && n >= 4)
      hipLaunchKernelGGL(extract_patches_kernel<true>, dim3(grid_for_ep(n / 4)), dim3(256), 0, s, x, o, (uint32_t)n, (uint32_t)r0,
                         FastDiv((uint32_t)feat), FastDiv((uint32_t)(k * C)), FastDiv((uint32_t)ow), FastDiv((uint32_t)oh), H, W, C, k, st, pt, pl);
    else
      hipLaunchKernelGGL(extract_patches_kernel<false>, dim3(grid_for_ep(n)), dim3(256), 0, s, x, o, (uint32_t)n, (uint32_t)r0,
                         FastDiv((uint32_t)feat), FastDiv((uint32_t)(k * C)), FastDiv((uint32_t)ow), FastDiv((uint32_t)oh), H, W, C, k, st, pt, pl);
  }
}
void launch_extract_patches_bwd(const float* dout, float* dx, int b, int H, int W, int C, int k, int st, hipStream_t s) {
  int oh, ow, pt, pl;
  extract_patches_geometry(H, W, k, st, &oh, &ow, &pt, &pl);
  const int64_t pixels = (int64_t)b * H * W;
  if (pixels == 0) return;
  const int64_t pix_per = std::max<int64_t>(1, (1ll << 30) / C);
  for (int64_t p0 = 0; p0 < pixels; p0 += pix_per) {
    const int64_t np = std::min(pix_per, pixels - p0), n = np * C;
    hipLaunchKernelGGL(extract_patches_bwd_kernel, dim3(grid_for_ep(n)), dim3(256), 0, s, dout, dx + p0 * C, (uint32_t)n, (uint32_t)p0,
                       FastDiv((uint32_t)C), FastDiv((uint32_t)W), FastDiv((uint32_t)H), FastDiv((uint32_t)st), k, st, oh, ow, pt, pl);
  }
}
