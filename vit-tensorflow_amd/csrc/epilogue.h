// GEMM epilogues shared by the generic (fp32-FMA) and the MFMA bf16 GEMM kernels.
// Each epilogue consumes 4 consecutive output columns of one row (float4 of fp32 accumulators).
#pragma once
#include "common.h"

enum EpiMode {
  EPI_STORE = 0,      // out[T] = alpha*acc (+bias)                                  QKV, dgrads
  EPI_STORE_F32 = 1,  // out[f32] = alpha*acc (+bias)                                logits, scores
  EPI_BIAS_GELU = 2,  // h = acc+bias; out2[T] = gelu(h); out[T] = h (T = float) or gelu'(h) (T = bf16: what the backward needs)   fc1 (vit.py:39,34)
  EPI_BIAS_RESID = 3, // f = acc+bias; out2[T] = f (iff scale); out[f32] = resid + f*scale   to_out / fc2 + residual (vit.py:101-102, cait.py:47-48)
  EPI_PATCH = 4,      // out[f32][img*ntok + tok_off + t] = acc + bias + pos[tok_off+t]   patch embed (vit.py:143,164-165)
  EPI_GELU_BWD = 5,   // out[T] = acc * gelu'(aux[T]) (T = float: aux = h) or acc * aux[T] (T = bf16: aux = the stored gelu'(h))   fc2 dgrad
  EPI_PARTIAL = 6,    // out[f32][z][row][col] = acc                                  split-K partial sums
};

// bf16 (throughput) mode: the fc1 epilogue stores gelu'(h) where parity mode stores h -- the only consumer of that buffer is the
// fc2 input-gradient epilogue, which then multiplies instead of re-evaluating the derivative (same bytes, ~40 VALU slots less per
// 4 elements in the backward epilogue; the forward pays one exp2 and three FMAs per element on top of the GELU it evaluates anyway)
template <typename T> inline constexpr bool kStoreGeluGrad = false;
template <> inline constexpr bool kStoreGeluGrad<bf16_t> = true;

struct EpiParams {
  void* out = nullptr;
  void* out2 = nullptr;
  const float* bias = nullptr;
  const float* resid = nullptr;
  const float* scale = nullptr;
  const void* aux = nullptr;
  const float* pos = nullptr;
  int64_t ldo = 0, ldo2 = 0, ldr = 0, ldaux = 0;
  int64_t out_batch_stride = 0, out_head_stride = 0;  // generic batched kernel only
  int64_t partial_stride = 0;                         // EPI_PARTIAL: elements per split slice
  float* colsum = nullptr;                            // EPI_GELU_BWD, LDS-staged bf16 kernels: [tile_m][N] column sums of the output
  int64_t ldcs = 0;                                   //   (bias gradient without re-reading the output), row = the launch's M-tile index
  int M = 0, N = 0;        // valid extents (rows >= M are written as zero for T outputs, skipped for f32)
  int np = 1, ntok = 1, tok_off = 0;
  int vec_ok = 1;          // 0: some pointer / leading dimension is not 16-B friendly -> scalar accesses
  int wide_ok = 0;         // 1: bf16 outputs / aux rows start 16-B aligned (leading dimensions % 8 == 0): eight-column epilogue form allowed
  int nt_out = 0;          // EPI_BIAS_GELU / EPI_STORE (bf16): 1 = `out` is written with non-temporal stores.  For tensors nothing reads soon
                           //   (gelu' of the fc1 epilogue: read by the backward; d(y) of the fc1 / qkv dgrad: read after the next weight gradient)
                           //   so that they do not push what the NEXT kernel reads out of the 256 MB memory-side cache (-0.3 and -0.13 ms per step)
  int zero_pad = 0;        // 1: rows in [M, tile end) of T outputs are written as zeros (buffers are row-padded)
  float alpha = 1.0f;
};

// T = storage type of "T" outputs / aux.  (row, col) are global tile coordinates; v holds columns col..col+3.
// All leading dimensions are multiples of 4 and col is a multiple of 4, so vector accesses are aligned.
template <int MODE, typename T>
__device__ __forceinline__ float4 epilogue_apply4(const EpiParams& p, int row, int col, float4 v, int64_t out_off = 0) {
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col >= p.N) return zero4;
  const bool full = (col + 3 < p.N) && p.vec_ok;
  float a[4] = {v.x * p.alpha, v.y * p.alpha, v.z * p.alpha, v.w * p.alpha};
  const bool row_ok = row < p.M;
  if (!row_ok && !p.zero_pad) return zero4;

  if (MODE == EPI_PARTIAL) {
    if (!row_ok) return zero4;
    float* o = (float*)p.out + out_off + (int64_t)row * p.ldo + col;
    if (full) *(float4*)o = make_float4(a[0], a[1], a[2], a[3]);
    else for (int i = 0; i < 4 && col + i < p.N; ++i) o[i] = a[i];
    return zero4;
  }
  if (p.bias != nullptr && MODE != EPI_GELU_BWD) {
    if (full) { float4 b = *(const float4*)(p.bias + col); a[0] += b.x; a[1] += b.y; a[2] += b.z; a[3] += b.w; }
    else for (int i = 0; i < 4 && col + i < p.N; ++i) a[i] += p.bias[col + i];
  }
  if (MODE == EPI_STORE) {
    T* o = (T*)p.out + out_off + (int64_t)row * p.ldo + col;
    if (!row_ok) { a[0] = a[1] = a[2] = a[3] = 0.f; }
    if (full) st4<T>(o, make_float4(a[0], a[1], a[2], a[3]));
    else for (int i = 0; i < 4 && col + i < p.N; ++i) stf<T>(o + i, a[i]);
  } else if (MODE == EPI_STORE_F32) {
    if (!row_ok) return zero4;
    float* o = (float*)p.out + out_off + (int64_t)row * p.ldo + col;
    if (full) *(float4*)o = make_float4(a[0], a[1], a[2], a[3]);
    else for (int i = 0; i < 4 && col + i < p.N; ++i) o[i] = a[i];
  } else if (MODE == EPI_BIAS_GELU) {
    T* o = (T*)p.out + (int64_t)row * p.ldo + col;
    T* o2 = (T*)p.out2 + (int64_t)row * p.ldo2 + col;
    float g[4];
    for (int i = 0; i < 4; ++i) {
      if (!row_ok) a[i] = 0.f;
      // GELU is evaluated on the value as stored (T-rounded) so that backward's gelu'(hpre) matches
      float hs = (float)(T)a[i];
      g[i] = row_ok ? gelu_t<T>(hs) : 0.f;
      if (kStoreGeluGrad<T>) a[i] = row_ok ? gelu_grad_t<T>(hs) : 0.f;   // bf16 mode keeps gelu'(h) instead of h (see EPI_GELU_BWD)
    }
    if (full) { st4<T>(o, make_float4(a[0], a[1], a[2], a[3])); st4<T>(o2, make_float4(g[0], g[1], g[2], g[3])); }
    else for (int i = 0; i < 4 && col + i < p.N; ++i) { stf<T>(o + i, a[i]); stf<T>(o2 + i, g[i]); }
  } else if (MODE == EPI_BIAS_RESID) {
    if (p.scale != nullptr && p.out2 != nullptr) {   // LayerScale: keep f(x) for dscale = sum g*f(x)
      T* o2 = (T*)p.out2 + (int64_t)row * p.ldo2 + col;
      float z[4] = {row_ok ? a[0] : 0.f, row_ok ? a[1] : 0.f, row_ok ? a[2] : 0.f, row_ok ? a[3] : 0.f};
      if (full) st4<T>(o2, make_float4(z[0], z[1], z[2], z[3]));
      else for (int i = 0; i < 4 && col + i < p.N; ++i) stf<T>(o2 + i, z[i]);
    }
    if (!row_ok) return zero4;
    float* o = (float*)p.out + (int64_t)row * p.ldo + col;
    const float* r = p.resid + (int64_t)row * p.ldr + col;
    if (full) {
      float4 rv = *(const float4*)r;
      float4 s = p.scale ? *(const float4*)(p.scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
      *(float4*)o = make_float4(rv.x + a[0] * s.x, rv.y + a[1] * s.y, rv.z + a[2] * s.z, rv.w + a[3] * s.w);
    } else {
      for (int i = 0; i < 4 && col + i < p.N; ++i) o[i] = r[i] + a[i] * (p.scale ? p.scale[col + i] : 1.f);
    }
  } else if (MODE == EPI_PATCH) {
    if (!row_ok) return zero4;
    const int img = row / p.np, t = row - img * p.np;
    const int64_t orow = (int64_t)img * p.ntok + p.tok_off + t;
    float* o = (float*)p.out + orow * p.ldo + col;
    const float* ps = p.pos + (int64_t)(p.tok_off + t) * p.ldr + col;
    if (full) { float4 q = *(const float4*)ps; *(float4*)o = make_float4(a[0] + q.x, a[1] + q.y, a[2] + q.z, a[3] + q.w); }
    else for (int i = 0; i < 4 && col + i < p.N; ++i) o[i] = a[i] + ps[i];
  } else if (MODE == EPI_GELU_BWD) {
    T* o = (T*)p.out + (int64_t)row * p.ldo + col;
    const T* h = (const T*)p.aux + (int64_t)row * p.ldaux + col;
    float g[4];
    if (full) {
      float4 hv = ld4<T>(h);
      if (kStoreGeluGrad<T>) { g[0] = a[0] * hv.x; g[1] = a[1] * hv.y; g[2] = a[2] * hv.z; g[3] = a[3] * hv.w; }
      else {
      g[0] = a[0] * gelu_grad_t<T>(hv.x); g[1] = a[1] * gelu_grad_t<T>(hv.y);
      g[2] = a[2] * gelu_grad_t<T>(hv.z); g[3] = a[3] * gelu_grad_t<T>(hv.w);
      }
      if (!row_ok) g[0] = g[1] = g[2] = g[3] = 0.f;
      st4<T>(o, make_float4(g[0], g[1], g[2], g[3]));
    } else {
      g[0] = g[1] = g[2] = g[3] = 0.f;
      for (int i = 0; i < 4 && col + i < p.N; ++i) {
        const float hv = ldf<T>(h + i);
        g[i] = row_ok ? a[i] * (kStoreGeluGrad<T> ? hv : gelu_grad_t<T>(hv)) : 0.f;
        stf<T>(o + i, g[i]);
      }
    }
    // what was stored (rounded to T): the fused column sums add exactly what a later pass over the output would read
    return make_float4((float)(T)g[0], (float)(T)g[1], (float)(T)g[2], (float)(T)g[3]);
  }
  return zero4;
}

// Branch-free form for INTERIOR tiles (every row < M, every column < N, vector-friendly pointers, alpha == 1): the caller has
// hoisted all those wave-uniform tests out of the per-element code, so the compiler can batch the loads and stores of a
// whole tile (the generic form above waits for each load before issuing the next: one memory operation in flight per wave).
// b4 / s4 = bias / LayerScale values of columns col..col+3 (loaded once per column group by the caller).
// The global READ an epilogue needs (fp32 residual / saved pre-activation) is split from the rest so that callers can issue
// a batch of them before the first store (the compiler cannot hoist loads over stores to possibly-aliasing pointers).
template <int MODE, typename T>
__device__ __forceinline__ float4 epilogue_fast_load(const EpiParams& p, int row, int col) {
  if (MODE == EPI_BIAS_RESID) return *(const float4*)(p.resid + (int64_t)row * p.ldr + col);
  if (MODE == EPI_GELU_BWD) return ld4<T>((const T*)p.aux + (int64_t)row * p.ldaux + col);
  return make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int MODE, typename T, bool HAS_BIAS, bool HAS_SCALE>
__device__ __forceinline__ float4 epilogue_fast4(const EpiParams& p, int row, int col, float4 v, float4 b4, float4 s4, float4 x, int64_t out_off) {
  if (HAS_BIAS && MODE != EPI_GELU_BWD && MODE != EPI_PARTIAL) { v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
  if (MODE == EPI_STORE) {
    if constexpr (sizeof(T) == 2) {
      if (p.nt_out) {
        bf16x4 o4;
        o4[0] = (bf16_t)v.x; o4[1] = (bf16_t)v.y; o4[2] = (bf16_t)v.z; o4[3] = (bf16_t)v.w;
        __builtin_nontemporal_store(o4, (bf16x4*)((bf16_t*)p.out + out_off + (int64_t)row * p.ldo + col));
        return v;
      }
    }
    st4<T>((T*)p.out + out_off + (int64_t)row * p.ldo + col, v);
  } else if (MODE == EPI_STORE_F32 || MODE == EPI_PARTIAL) {
    *(float4*)((float*)p.out + out_off + (int64_t)row * p.ldo + col) = v;
  } else if (MODE == EPI_BIAS_GELU) {
    const float4 hs = make_float4((float)(T)v.x, (float)(T)v.y, (float)(T)v.z, (float)(T)v.w);   // GELU of the value as T would store it
    if constexpr (kStoreGeluGrad<T>) {
      float4 g, gd;
      gelu_both4(hs, g, gd);
      if (p.nt_out) {
        bf16x4 o4;
        o4[0] = (bf16_t)gd.x; o4[1] = (bf16_t)gd.y; o4[2] = (bf16_t)gd.z; o4[3] = (bf16_t)gd.w;
        __builtin_nontemporal_store(o4, (bf16x4*)((bf16_t*)p.out + (int64_t)row * p.ldo + col));
      } else {
        st4<T>((T*)p.out + (int64_t)row * p.ldo + col, gd);
      }
      st4<T>((T*)p.out2 + (int64_t)row * p.ldo2 + col, g);
    } else {
      st4<T>((T*)p.out + (int64_t)row * p.ldo + col, v);
      st4<T>((T*)p.out2 + (int64_t)row * p.ldo2 + col, gelu4_t<T>(hs));
    }
  } else if (MODE == EPI_BIAS_RESID) {
    if (HAS_SCALE) {
      if (p.out2) st4<T>((T*)p.out2 + (int64_t)row * p.ldo2 + col, v);
      v.x *= s4.x; v.y *= s4.y; v.z *= s4.z; v.w *= s4.w;
    }
    *(float4*)((float*)p.out + (int64_t)row * p.ldo + col) = make_float4(x.x + v.x, x.y + v.y, x.z + v.z, x.w + v.w);
  } else if (MODE == EPI_GELU_BWD) {
    const float4 gd = kStoreGeluGrad<T> ? x : gelu_grad4_t<T>(x);
    const float4 gq = make_float4(v.x * gd.x, v.y * gd.y, v.z * gd.z, v.w * gd.w);
    st4<T>((T*)p.out + (int64_t)row * p.ldo + col, gq);
    return make_float4((float)(T)gq.x, (float)(T)gq.y, (float)(T)gq.z, (float)(T)gq.w);   // as stored (see epilogue_apply4)
  }
  return v;
}
// Eight consecutive columns per lane for the bf16-output epilogues (EPI_STORE, EPI_BIAS_GELU, EPI_GELU_BWD): ONE 16-byte global
// access per lane and output row instead of two 8-byte ones.  The bf16 store tail of an MFMA epilogue is bound by the number of
// store instructions, not by their bytes (MI355X: a 16 x dwordx2 tail takes twice as long as the same bytes as 8 x dwordx4), so
// halving the instruction count is what shortens it.  Interior tiles only; same arithmetic as epilogue_fast4 on each half.
__device__ __forceinline__ bf16x8 pack_bf16x8(float4 a, float4 b) {
  bf16x8 o;
  o[0] = (bf16_t)a.x; o[1] = (bf16_t)a.y; o[2] = (bf16_t)a.z; o[3] = (bf16_t)a.w;
  o[4] = (bf16_t)b.x; o[5] = (bf16_t)b.y; o[6] = (bf16_t)b.z; o[7] = (bf16_t)b.w;
  return o;
}
template <int MODE>
__device__ __forceinline__ bf16x8 epilogue_wide_load(const EpiParams& p, int row, int col) {
  if (MODE == EPI_GELU_BWD) return *(const bf16x8*)((const bf16_t*)p.aux + (int64_t)row * p.ldaux + col);
  return bf16x8{};
}
// returns (as two float4 through lo / hi) what was stored, rounded to bf16 -- the fused column sums of EPI_GELU_BWD
template <int MODE, bool HAS_BIAS>
__device__ __forceinline__ void epilogue_wide8(const EpiParams& p, int row, int col, float4& lo, float4& hi, float4 b_lo, float4 b_hi, bf16x8 x,
                                               int64_t out_off) {
  if (HAS_BIAS && MODE != EPI_GELU_BWD) {
    lo.x += b_lo.x; lo.y += b_lo.y; lo.z += b_lo.z; lo.w += b_lo.w;
    hi.x += b_hi.x; hi.y += b_hi.y; hi.z += b_hi.z; hi.w += b_hi.w;
  }
  if (MODE == EPI_STORE) {
    if (p.nt_out) __builtin_nontemporal_store(pack_bf16x8(lo, hi), (bf16x8*)((bf16_t*)p.out + out_off + (int64_t)row * p.ldo + col));
    else *(bf16x8*)((bf16_t*)p.out + out_off + (int64_t)row * p.ldo + col) = pack_bf16x8(lo, hi);
  } else if (MODE == EPI_BIAS_GELU) {
    const bf16x8 h = pack_bf16x8(lo, hi);       // GELU and its derivative of the pre-activation as bf16 would store it
    float4 g0, d0, g1, d1;
    gelu_both4(make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]), g0, d0);
    gelu_both4(make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]), g1, d1);
    if (p.nt_out) __builtin_nontemporal_store(pack_bf16x8(d0, d1), (bf16x8*)((bf16_t*)p.out + (int64_t)row * p.ldo + col));
    else *(bf16x8*)((bf16_t*)p.out + (int64_t)row * p.ldo + col) = pack_bf16x8(d0, d1);
    *(bf16x8*)((bf16_t*)p.out2 + (int64_t)row * p.ldo2 + col) = pack_bf16x8(g0, g1);
  } else if (MODE == EPI_GELU_BWD) {
    lo = make_float4(lo.x * (float)x[0], lo.y * (float)x[1], lo.z * (float)x[2], lo.w * (float)x[3]);
    hi = make_float4(hi.x * (float)x[4], hi.y * (float)x[5], hi.z * (float)x[6], hi.w * (float)x[7]);
    const bf16x8 o = pack_bf16x8(lo, hi);
    *(bf16x8*)((bf16_t*)p.out + (int64_t)row * p.ldo + col) = o;
    lo = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
    hi = make_float4((float)o[4], (float)o[5], (float)o[6], (float)o[7]);
  }
}
__device__ __forceinline__ bool epilogue_fast_ok(const EpiParams& p, int mode) {
  return p.vec_ok && p.alpha == 1.0f && mode != EPI_PATCH;
}
