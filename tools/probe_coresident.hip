// Hardware probe (round 6): can an HBM-bound kernel run ON THE SAME CUs as a resident persistent GEMM, in the registers the GEMM leaves free?
// The step is a serial sum of an MFMA-bound part and an HBM-bound part: the GEMM workgroups (8 waves x ~200-250 VGPRs, 128 KiB
// of LDS) fill a CU's register file, so a LayerNorm launched beside a weight gradient waits for it.  A SIMD has 512 VGPRs per lane; two GEMM waves of
// 208 leave 96, of 200 leave 112.  This probe runs
//   G<R>: a GEMM-like persistent kernel -- 256 workgroups x 512 threads, 128 KiB of LDS, R VGPRs per wave, an MFMA loop on registers (no memory traffic)
//   S<R>: a streaming kernel -- 256-thread workgroups, R VGPRs per wave, no LDS, float4 read + write of 2 x 302 MB (HBM-bound)
// alone and together on two streams, and prints the three times: together ~ max(alone) means co-residency, ~ sum means time-slicing.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_coresident.hip -o tools/probe_coresident
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GEMM_KERNEL(N, REG)                                                                                                   \
  __global__ __launch_bounds__(512) void gemm_like_##N(float* out, int iters) {                                                \
    extern __shared__ char smem[];                                                                                            \
    asm volatile("v_mov_b32 " REG ", 0" ::: REG);                                                                             \
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};                                                                                \
    bf16x8 x, y;                                                                                                              \
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(i * 0.5f); }                    \
    if (threadIdx.x == 0) smem[0] = 1;                                                                                        \
    for (int it = 0; it < iters; ++it) {                                                                                      \
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);                                                        \
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);                                                        \
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);                                                        \
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);                                                        \
    }                                                                                                                         \
    float s = 0.f;                                                                                                            \
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];                                                          \
    if (s == 12345.678f) out[blockIdx.x] = s;                                                                                 \
  }
GEMM_KERNEL(200, "v199")
GEMM_KERNEL(208, "v207")
GEMM_KERNEL(240, "v239")
GEMM_KERNEL(64, "v63")
// control: the same residency (200 VGPRs, 8 waves, 128 KiB of LDS) with the waves ASLEEP instead of issuing MFMAs -- tells residency from issue starvation
__global__ __launch_bounds__(512) void sleeper_200(float* out, int iters) {
  extern __shared__ char smem[];
  asm volatile("v_mov_b32 v199, 0" ::: "v199");
  if (threadIdx.x == 0) smem[0] = 1;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 35000) __builtin_amdgcn_s_sleep(32);   // 100 MHz clock: 350 us
  if (iters == -1) out[blockIdx.x] = 1.f;
}

#define STREAM_KERNEL(N, REG)                                                                                                 \
  __global__ __launch_bounds__(256) void stream_##N(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {      \
    asm volatile("v_mov_b32 " REG ", 0" ::: REG);                                                                             \
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {                                 \
      float4 v = src[i];                                                                                                      \
      v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;                                                                         \
      dst[i] = v;                                                                                                             \
    }                                                                                                                         \
  }
// the streaming kernel at the HIGHEST wave priority: its few instructions win the issue arbitration against the resident MFMA waves
#define STREAM_PRIO_KERNEL(N, REG)                                                                                            \
  __global__ __launch_bounds__(256) void stream_prio_##N(const float4* __restrict__ src, float4* __restrict__ dst, long n4) { \
    __builtin_amdgcn_s_setprio(3);                                                                                            \
    asm volatile("v_mov_b32 " REG ", 0" ::: REG);                                                                             \
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {                                 \
      float4 v = src[i];                                                                                                      \
      v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;                                                                         \
      dst[i] = v;                                                                                                             \
    }                                                                                                                         \
  }
STREAM_PRIO_KERNEL(32, "v31")
STREAM_PRIO_KERNEL(96, "v95")
STREAM_KERNEL(32, "v31")
STREAM_KERNEL(96, "v95")
STREAM_KERNEL(112, "v111")
STREAM_KERNEL(128, "v127")

template <typename KG, typename KS>
int run(const char* what, KG kg, KS ks, float* out, const float4* src, float4* dst, long n4, int iters, hipStream_t sa, hipStream_t sb, int LDS = 128 * 1024, int gwg = 256) {
  CK(hipFuncSetAttribute((const void*)kg, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  auto G = [&]() { hipLaunchKernelGGL(kg, dim3(gwg), dim3(512), LDS, sa, out, iters); };
  auto S = [&]() { hipLaunchKernelGGL(ks, dim3(4096), dim3(256), 0, sb, src, dst, n4); };
  float tg = 0, ts = 0, tb = 1e30f;
  G(); S(); CK(hipDeviceSynchronize());   // warm-up
  CK(hipEventRecord(e0, sa)); G(); CK(hipEventRecord(e1, sa)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tg, e0, e1));
  CK(hipEventRecord(e0, sb)); S(); CK(hipEventRecord(e1, sb)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ts, e0, e1));
  float tg_in = 0, ts_in = 0;
  for (int rep = 0; rep < 3; ++rep) {
    // together: the GEMM-like kernel first (resident on every CU), the streaming kernel 20 us later on the other stream
    CK(hipEventRecord(e0, sa));
    G();
    CK(hipEventRecord(e1, sa));
    CK(hipStreamWaitEvent(sb, e0, 0));
    hipEvent_t s0, s1;
    CK(hipEventCreate(&s0)); CK(hipEventCreate(&s1));
    CK(hipEventRecord(s0, sb)); S(); CK(hipEventRecord(s1, sb));
    CK(hipEventRecord(e2, sb));
    CK(hipDeviceSynchronize());
    float a = 0, b = 0, g1 = 0, s1ms = 0;
    CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e0, e2));
    CK(hipEventElapsedTime(&g1, e0, e1)); CK(hipEventElapsedTime(&s1ms, s0, s1));
    const float both = a > b ? a : b;
    if (both < tb) { tb = both; tg_in = g1; ts_in = s1ms; }
    CK(hipEventDestroy(s0)); CK(hipEventDestroy(s1));
  }
  printf("%-34s gemm alone %7.1f us  stream alone %7.1f us  sum %7.1f | together %7.1f us (gemm %7.1f, stream %7.1f)  -> %s\n", what, tg * 1e3, ts * 1e3,
         (tg + ts) * 1e3, tb * 1e3, tg_in * 1e3, ts_in * 1e3, tb < 0.8f * (tg + ts) ? "OVERLAP" : "serial");
  return 0;
}

int main(int argc, char** argv) {
  const int same_prio = argc > 1 ? atoi(argv[1]) : 0;
  const long n4 = 302l * 1024 * 1024 / 16;   // 302 MB each way
  float4 *src, *dst; float* out;
  CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(src, 0, n4 * 16));
  hipStream_t sa, sb;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  if (same_prio == 1) { CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
  else if (same_prio == 2) { CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo)); }
  else {
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo));   // the GEMM at the lowest priority (as the engine's weight-gradient stream)
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
  }
  printf("stream priorities: mode %d (0 = gemm low / stream high, 1 = both default, 2 = gemm high / stream low)\n", same_prio);
  const int iters = 3000;   // 4 MFMAs x 3000 x 32 cycles on one SIMD with two waves: ~2 x 160 us at 2.4 GHz
  run("gemm 200 VGPRs + stream 32", gemm_like_200, stream_32, out, src, dst, n4, iters, sa, sb);
  run("gemm 200 VGPRs + stream 96", gemm_like_200, stream_96, out, src, dst, n4, iters, sa, sb);
  run("gemm 200 VGPRs + stream 112", gemm_like_200, stream_112, out, src, dst, n4, iters, sa, sb);
  run("gemm 200 VGPRs + stream 128", gemm_like_200, stream_128, out, src, dst, n4, iters, sa, sb);
  run("gemm 208 VGPRs + stream 32", gemm_like_208, stream_32, out, src, dst, n4, iters, sa, sb);
  run("gemm 208 VGPRs + stream 96", gemm_like_208, stream_96, out, src, dst, n4, iters, sa, sb);
  run("gemm 208 VGPRs + stream 112", gemm_like_208, stream_112, out, src, dst, n4, iters, sa, sb);
  run("gemm 240 VGPRs + stream 32", gemm_like_240, stream_32, out, src, dst, n4, iters, sa, sb);
  run("gemm 240 VGPRs + stream 96", gemm_like_240, stream_96, out, src, dst, n4, iters, sa, sb);
  // controls: a small GEMM-like kernel (64 VGPRs, 32 KiB of LDS: plenty of room beside it), and the big one on HALF the CUs
  run("gemm 64 VGPRs 32 KiB + stream 32", gemm_like_64, stream_32, out, src, dst, n4, iters, sa, sb, 32 * 1024);
  run("gemm 64 VGPRs 32 KiB + stream 96", gemm_like_64, stream_96, out, src, dst, n4, iters, sa, sb, 32 * 1024);
  run("gemm 200 VGPRs, 128 WGs + stream 32", gemm_like_200, stream_32, out, src, dst, n4, iters, sa, sb, 128 * 1024, 128);
  run("gemm 200 VGPRs, 64 KiB + stream 32", gemm_like_200, stream_32, out, src, dst, n4, iters, sa, sb, 64 * 1024);
  run("gemm 200 VGPRs + PRIO-3 stream 32", gemm_like_200, stream_prio_32, out, src, dst, n4, iters, sa, sb);
  run("gemm 200 VGPRs + PRIO-3 stream 96", gemm_like_200, stream_prio_96, out, src, dst, n4, iters, sa, sb);
  run("gemm 208 VGPRs + PRIO-3 stream 96", gemm_like_208, stream_prio_96, out, src, dst, n4, iters, sa, sb);
  run("SLEEPING 200 VGPRs 128 KiB + stream 32", sleeper_200, stream_32, out, src, dst, n4, iters, sa, sb);
  run("SLEEPING 200 VGPRs 128 KiB + stream 96", sleeper_200, stream_96, out, src, dst, n4, iters, sa, sb);
  run("SLEEPING 200 VGPRs 128 KiB + stream 128", sleeper_200, stream_128, out, src, dst, n4, iters, sa, sb);
  run("gemm 200 VGPRs, 1 KiB + stream 32", gemm_like_200, stream_32, out, src, dst, n4, iters, sa, sb, 1024);
  return 0;
}
