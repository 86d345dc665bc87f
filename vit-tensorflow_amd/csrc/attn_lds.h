// LDS images and MFMA fragment helpers shared by the fused attention kernels (attn_bf16.hip: vit.py:73-82;
// attn_deepvit_fused.hip: deepvit.py:73-91): one head's [rows][64] bf16 matrix as a swizzled row-major LDS image, staged with
// direct-to-LDS loads, read either as row fragments (ds_read_b128) or through the hardware transpose (ds_read_b64_tr_b16).
#pragma once
#include "kernels.h"

namespace attn_lds {

constexpr int DH = 64;
constexpr int ROWB = DH * 2;  // 128-byte rows

#ifdef VITX_ATTN_PROBE_NOEXP   // timing switch of tools/probe_attn (wrong results): what do the transcendentals cost?
__device__ __forceinline__ float fast_exp2(float x) { return x; }
#else
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // raw v_exp_f32 (inputs are <= 0 or masked)
#endif

// 16-B chunk swizzle of the 128-B rows.  The key moves in steps of TWO chunks (32 B) per row pair: the transpose reads
// (ds_read_b64_tr_b16, 32-lane service groups touching 8 rows x 32 B) need rows r and r+2 in different 32-B bank groups -- with a
// one-chunk step they landed on the same 8 banks (2-way conflict on every transpose read: a quarter of all LDS cycles in these
// kernels) -- and the row-major ds_read_b128 fragments stay conflict-free (16 distinct (row parity, chunk) pairs per 16-lane group).
__device__ __forceinline__ int swz_key(int row) { return ((row >> 1) & 3) << 1; }
__device__ __forceinline__ int swz_chunk(int row, int chunk) { return (chunk ^ swz_key(row)) << 4; }

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.f;
  return z;
}

// Asynchronous staging of one head's [npad][64] bf16 matrix into the swizzled row-major LDS image with direct-to-LDS loads
// (global_load_lds_dwordx4: 1 KiB = 8 rows per wave instruction, no VGPR round trip, every load of the workgroup in flight at
// once).  The LDS image is lane-linear, so the swizzle goes on the per-lane SOURCE address; rows >= nvalid read a zero page.
// Caller: s_waitcnt vmcnt(0) + barrier before the first ds_read.
__device__ __forceinline__ void stage_head_dma(const bf16_t* src, int64_t stride, int nvalid, int npad, char* rm, const bf16_t* zero_page,
                                               int wave, int lane, int nwaves) {
  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef const __attribute__((address_space(1))) void gbl_void_t;
  for (int i = wave; i < npad / 8; i += nwaves) {
    const int row = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_key(row);
    const bf16_t* p = row < nvalid ? src + (int64_t)row * stride + c * 8 : zero_page + (lane & 7) * 8;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)p, (lds_void_t*)(rm + i * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 frag_rm(const char* rm, int row, int chunk) {
  return *(const bf16x8*)(rm + row * ROWB + swz_chunk(row, chunk));
}
// Transposed operand straight from the row-major swizzled image via the LDS hardware transpose read
// (ds_read_b64_tr_b16): fragment row = feature 16c + (lane&15), k-slots (g,e): e<4 -> row 32u+4g+e, e>=4 -> row 32u+16+4g+(e-4)
// of the image.  Within a 16-lane group, lanes 4j..4j+3 address 16 consecutive features of image row (base + j) and lane q
// receives feature q of rows base..base+3.
__device__ __forceinline__ bf16x8 frag_trr(const char* rm, int c, int u, int lane) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int q = lane & 15, g = lane >> 4;
  const int row0 = 32 * u + 4 * g + (q >> 2), row1 = row0 + 16;
  const int b = 32 * c + 8 * (q & 3);
  const int a0 = row0 * ROWB + ((((b >> 4) ^ swz_key(row0)) << 4) | (b & 15));
  const int a1 = row1 * ROWB + ((((b >> 4) ^ swz_key(row1)) << 4) | (b & 15));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(rm + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(rm + a1));
  union { struct { s16x4 a, b; } s; bf16x8 v; } uu;
  uu.s.a = lo; uu.s.b = hi;
  return uu.v;
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = (bf16_t)a[e]; r[4 + e] = (bf16_t)b[e]; }
  return r;
}
#ifdef VITX_ATTN_PROBE_NOMFMA  // timing switch of tools/probe_attn (wrong results): one VALU op per MFMA, operands kept alive
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  f32x4 r = c;
  r[0] += (float)a[0] * (float)b[0];
  return r;
}
#else
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#endif


}  // namespace attn_lds
