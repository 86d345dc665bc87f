// Shared device/host helpers for libvitx (gfx950 only; wave = 64).
#pragma once
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

// Fast variant for the bf16 throughput mode (Abramowitz-Stegun 7.1.26, |erf error| <= 1.5e-7 -- far below bf16 resolution):
// the branchy library erff (~60 VALU ops) made the GELU epilogues VALU-bound; this form is 13 ops, two of them transcendental.
//   h(x) = 0.5 * erfc(|x| / sqrt2) = t * P(t) * exp(-x^2/2),  t = 1 / (1 + p |x| / sqrt2)      (0.5 folded into P's coefficients)
//   Phi(x) = x >= 0 ? 1 - h : h        gelu(x) = x Phi(x) = max(x, 0) - |x| h        gelu'(x) = Phi(x) + x exp(-x^2/2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_half_erfc(float x, float& e) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
  e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  return p * t * e;
}
__device__ __forceinline__ float gelu_phi_fast(float x, float& e) {
  const float h = gelu_half_erfc(x, e);
  return x >= 0.f ? 1.0f - h : h;
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) { return gelu_f(x); }
template <> __device__ __forceinline__ float gelu_t<bf16_t>(float x) {
  float e;
  const float h = gelu_half_erfc(x, e);
  return fmaf(-fabsf(x), h, fmaxf(x, 0.f));
}
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) { return gelu_grad_f(x); }
template <> __device__ __forceinline__ float gelu_grad_t<bf16_t>(float x) {
  float e;
  const float phi = gelu_phi_fast(x, e);
  return fmaf(x * 0.39894228040143267794f, e, phi);
}

// An SGPR zero the optimiser cannot see through.  Adding it to the mixing-matrix pointers INSIDE the row loop keeps the (wave-uniform,
// scalar) weight loads inside the loop: hoisted out of it they need 256-512 SGPRs at once and the allocator parks them in VGPR lanes
// (3208 v_readlane per row in the first version of cait_chain_fwd_kernel -- 147 us per launch instead of ~60).
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
