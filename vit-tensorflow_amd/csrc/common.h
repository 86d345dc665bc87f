// Shared device/host helpers for libvitx (gfx950 only; wave = 64).
#pragma once
#include "env.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return (float)(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = (bf16_t)v; }

// 4 consecutive elements (16-B aligned for float, 8-B aligned for bf16)
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
  bf16x4 v = *(const bf16x4*)p;
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *(float4*)p = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float4 v) {
  bf16x4 o;
  o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
  *(bf16x4*)p = o;
}

// Wave-wide reductions on the DPP path (no LDS traffic; the generic __shfl_xor lowers to ds_bpermute_b32, one LDS instruction per
// step -- 192 of them per row made the fused head-axis kernels LDS-bound).  All 64 lanes must be active.  The result is read from
// lane 63 into an SGPR, i.e. it is wave-uniform.
//   quad_perm [1,0,3,2] / [2,3,0,1]: xor 1 / xor 2 inside a quad; row_ror:4 / row_ror:8: rotate inside a 16-lane row;
//   row_bcast:15 (rows 1,3 += lane 15 of the previous row), row_bcast:31 (rows 2,3 += lane 31): gfx9 DPP controls.
#define VITX_DPP_STEP(OP, ctrl, rmask, ident)                                                                                         \
  v = OP(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(ident)), __builtin_bit_cast(int, v), \
                                                                    (ctrl), (rmask), 0xF, false)))
__device__ __forceinline__ float vitx_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum(float v) {
  VITX_DPP_STEP(vitx_addf, 0xB1, 0xF, 0.f);    // quad_perm [1,0,3,2]
  VITX_DPP_STEP(vitx_addf, 0x4E, 0xF, 0.f);    // quad_perm [2,3,0,1]
  VITX_DPP_STEP(vitx_addf, 0x124, 0xF, 0.f);   // row_ror:4
  VITX_DPP_STEP(vitx_addf, 0x128, 0xF, 0.f);   // row_ror:8  -> every lane holds its row's sum
  VITX_DPP_STEP(vitx_addf, 0x142, 0xA, 0.f);   // row_bcast:15
  VITX_DPP_STEP(vitx_addf, 0x143, 0xC, 0.f);   // row_bcast:31 -> lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  VITX_DPP_STEP(fmaxf, 0xB1, 0xF, -INFINITY);
  VITX_DPP_STEP(fmaxf, 0x4E, 0xF, -INFINITY);
  VITX_DPP_STEP(fmaxf, 0x124, 0xF, -INFINITY);
  VITX_DPP_STEP(fmaxf, 0x128, 0xF, -INFINITY);
  VITX_DPP_STEP(fmaxf, 0x142, 0xA, -INFINITY);
  VITX_DPP_STEP(fmaxf, 0x143, 0xC, -INFINITY);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef VITX_DPP_STEP

// exact-erf GELU (vit.py:34) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// Fast variants for the bf16 throughput mode.  The GELU epilogues are VALU-bound (measured with cycle stamps: 23.5k cycles per
// 256x256 tile whether 64 or 256 CUs run, against 9-12k for a plain bf16 store): a wave64 VALU op takes 4 cycles, a transcendental
// one 16, and every lane evaluates 128 elements per tile.  So no rcp / exp in the forward form, and everything written on 2-vectors
// so that it compiles to packed fp32 ops (v_pk_mul_f32 / v_pk_fma_f32: two elements per instruction):
//   erf(x / sqrt2) ~ x * P(u),  u = min(x^2 / 2, 7.84),  clamped to [-1, 1]     (degree-7 minimax fit in u on |x| <= 3.96:
//   |erf error| <= 4.2e-5, 7.5e-5 beyond the clamp -> |gelu error| <= 1.5e-4, far below the bf16 spacing of the stored activations)
//   gelu(x) = x/2 + x/2 erf        gelu'(x) = 1/2 + 1/2 erf + x exp(-x^2/2) / sqrt(2 pi)   (one hardware exp2)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ f32x2 vfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float vsplat(float, float v) { return v; }
__device__ __forceinline__ f32x2 vsplat(f32x2, float v) { return f32x2{v, v}; }
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ f32x2 vmin(f32x2 a, f32x2 b) { return f32x2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
__device__ __forceinline__ float vclamp1(float a) { return __builtin_amdgcn_fmed3f(a, -1.0f, 1.0f); }
__device__ __forceinline__ f32x2 vclamp1(f32x2 a) { return f32x2{__builtin_amdgcn_fmed3f(a.x, -1.0f, 1.0f), __builtin_amdgcn_fmed3f(a.y, -1.0f, 1.0f)}; }
__device__ __forceinline__ float vexp2(float a) { return __builtin_amdgcn_exp2f(a); }
__device__ __forceinline__ f32x2 vexp2(f32x2 a) { return f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
template <typename V> __device__ __forceinline__ V gelu_erf_fast(V x, V& u0) {   // erf(x / sqrt2); u0 = x^2 / 2
  u0 = x * x * vsplat(x, 0.5f);
  const V u = vmin(u0, vsplat(x, 7.84f));
  V p = vsplat(x, -4.356138932e-07f);
  p = vfma(p, u, vsplat(x, 1.643961321e-05f));
  p = vfma(p, u, vsplat(x, -2.719887553e-04f));
  p = vfma(p, u, vsplat(x, 2.634852515e-03f));
  p = vfma(p, u, vsplat(x, -1.693103523e-02f));
  p = vfma(p, u, vsplat(x, 7.756211587e-02f));
  p = vfma(p, u, vsplat(x, -2.648631880e-01f));
  p = vfma(p, u, vsplat(x, 7.977257765e-01f));
  return vclamp1(p * x);
}
template <typename V> __device__ __forceinline__ V gelu_fast(V x) {
  V u0;
  const V e = gelu_erf_fast(x, u0);
  const V hx = x * vsplat(x, 0.5f);
  return vfma(hx, e, hx);
}
template <typename V> __device__ __forceinline__ V gelu_grad_fast(V x) {
  V u0;
  const V e = gelu_erf_fast(x, u0);
  const V phi = vfma(e, vsplat(x, 0.5f), vsplat(x, 0.5f));
  const V ex = vexp2(u0 * vsplat(x, -1.44269504088896340736f));
  return vfma(x * vsplat(x, 0.39894228040143267794f), ex, phi);
}
// gelu(x) and gelu'(x) from ONE evaluation of the erf polynomial (the bf16 forward epilogue of fc1 stores both: the backward
// epilogue of the fc2 input gradient is then a single multiply by the stored derivative instead of a polynomial + exp2 per element)
template <typename V> __device__ __forceinline__ void gelu_both_fast(V x, V& g, V& gd) {
  V u0;
  const V e = gelu_erf_fast(x, u0);
  const V hx = x * vsplat(x, 0.5f);
  g = vfma(hx, e, hx);
  const V phi = vfma(e, vsplat(x, 0.5f), vsplat(x, 0.5f));
  const V ex = vexp2(u0 * vsplat(x, -1.44269504088896340736f));
  gd = vfma(x * vsplat(x, 0.39894228040143267794f), ex, phi);
}
__device__ __forceinline__ void gelu_both4(float4 x, float4& g, float4& gd) {
  f32x2 ga, gb, da, db;
  gelu_both_fast<f32x2>(f32x2{x.x, x.y}, ga, da);
  gelu_both_fast<f32x2>(f32x2{x.z, x.w}, gb, db);
  g = make_float4(ga.x, ga.y, gb.x, gb.y);
  gd = make_float4(da.x, da.y, db.x, db.y);
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) { return gelu_f(x); }
template <> __device__ __forceinline__ float gelu_t<bf16_t>(float x) { return gelu_fast<float>(x); }
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) { return gelu_grad_f(x); }
template <> __device__ __forceinline__ float gelu_grad_t<bf16_t>(float x) { return gelu_grad_fast<float>(x); }
// four elements at a time (the vectorised epilogues): two packed pairs in bf16 mode, the exact scalar form in parity mode
template <typename T> __device__ __forceinline__ float4 gelu4_t(float4 x) {
  return make_float4(gelu_f(x.x), gelu_f(x.y), gelu_f(x.z), gelu_f(x.w));
}
template <> __device__ __forceinline__ float4 gelu4_t<bf16_t>(float4 x) {
  const f32x2 a = gelu_fast<f32x2>(f32x2{x.x, x.y}), b = gelu_fast<f32x2>(f32x2{x.z, x.w});
  return make_float4(a.x, a.y, b.x, b.y);
}
template <typename T> __device__ __forceinline__ float4 gelu_grad4_t(float4 x) {
  return make_float4(gelu_grad_f(x.x), gelu_grad_f(x.y), gelu_grad_f(x.z), gelu_grad_f(x.w));
}
template <> __device__ __forceinline__ float4 gelu_grad4_t<bf16_t>(float4 x) {
  const f32x2 a = gelu_grad_fast<f32x2>(f32x2{x.x, x.y}), b = gelu_grad_fast<f32x2>(f32x2{x.z, x.w});
  return make_float4(a.x, a.y, b.x, b.y);
}

// An SGPR zero the optimiser cannot see through.  Adding it to the mixing-matrix pointers INSIDE the row loop keeps the (wave-uniform,
// scalar) weight loads inside the loop: hoisted out of it they need 256-512 SGPRs at once and the allocator parks them in VGPR lanes
// (3208 v_readlane per row in the first version of cait_chain_fwd_kernel -- 147 us per launch instead of ~60).
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}

// ---- LDS-DMA hidden from the compiler's s_waitcnt bookkeeping.
// hipcc (ROCm 7.2) treats every LDS-DMA builtin as a pending LDS write that any later ds_read may alias: in the pipelined GEMM loops it put
// `s_waitcnt vmcnt(0)` in front of the fragment reads of the k-step that follows each DMA piece (NT kernel: once per K-tile, weight-gradient
// kernel: before EVERY k-step), i.e. the MFMA waves waited for pieces they had issued a few hundred cycles earlier and the "prefetch" was
// drained three times per K-tile.  Issued from inline asm the pieces are invisible to that pass; the only waits left are the explicit
// `s_waitcnt vmcnt(N)` of the schedule (issuing wave) + the workgroup barrier in front of the first read of the buffer.  Ordinary loads and
// stores the compiler counts itself only become more conservative (hidden pieces add to the hardware counter, never to the compiler's).
//   dst  = wave-uniform LDS byte address (lane i lands at dst + 16 i);  rsrc = buffer resource of the operand (kernel-argument pointer);
//   voff = per-lane byte offset;  soff = wave-uniform byte offset.  M0 is written inside the statement and nothing else in these kernels uses M0
//   once the builtins are gone.  An "m0" entry in the clobber list does not make that a compiler guarantee: M0 is a reserved register to hipcc,
//   which answers "inline asm clobber list contains reserved registers: m0 ... may not be preserved across the asm statement" (-Winline-asm) and
//   generates the same code.  The guarantee is the ISA gate: build.py disassembles these kernels after every compile and fails the build if
//   anything but these statements writes M0 (tools/isa_check.py; tests/test_isa.py repeats it on the shipped objects).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 vitx_make_rsrc(const void* p) {
  const uint64_t a = (uint64_t)p;
  return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
}
__device__ __forceinline__ uint32_t vitx_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void vitx_dma16(i32x4 rsrc, uint32_t dst, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// Continuation piece: M0 as the previous vitx_dma16 of this wave left it, LDS destination = M0 + IMM (+ 16 lane), global address = ... + IMM as
// well (the instruction's offset field feeds both), so the caller passes voff - IMM.  Writing M0 for every piece serialises a wave's pieces
// on the M0 dependency (the next s_mov m0 waits until the address path has consumed the previous value); a wave's pieces of one operand are
// therefore laid out back to back in LDS (<= 4 KiB: the 12-bit offset field) and share one M0 value.  Nothing else in the kernels that use
// this touches M0 between the pieces of a group (checked in the ISA).
template <int IMM>
__device__ __forceinline__ void vitx_dma16_cont(i32x4 rsrc, uint32_t voff_minus_imm, uint32_t soff) {
  static_assert(IMM > 0 && IMM < 4096, "12-bit unsigned offset field");
  asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff_minus_imm), "s"(rsrc), "s"(soff), "i"(IMM) : "memory");
}

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting and one process may drive handles on several GPUs (a per-process
// `static bool` guard left the second device without it and the launch failed there): remembered per (device, kernel).
#include <mutex>
#include <set>
#include <utility>
inline void vitx_set_max_smem(const void* kern, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert({dev, kern}).second) (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
