// Hardware probe: LDS read bandwidth of ONE CU on gfx950 -- ds_read_b128 / ds_read_b64 / ds_read_b64_tr_b16 issued back to back by 4 / 8 / 16 waves with
// conflict-free addresses (the GEMM fragment pattern: 32 rows x 128 B, 16-B chunks swizzled), bytes per shader clock (s_memtime ticks) and per
// 100-MHz tick (s_memrealtime) so that the clock under this load can be read off as well.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: ds_read_b128, 1: ds_read_b64, 2: ds_read_b64_tr_b16
__global__ void probe(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)smem)[i] = (float)i;
  __syncthreads();
  // fragment addressing of the GEMM kernels: row = lane & 31 (128-B rows), 16-B chunk = (khalf ^ swizzle(row))
  const int row = lane & 31, kh = lane >> 5;
  const char* p = smem + (wave & 3) * 4096 + row * 128 + (((kh ^ ((row >> 1) & 7)) & 7) << 4);
  float acc = 0.f;
  const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)(char*)p;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned a = addr + ((u & 3) << 5) + (((u >> 2) & 3) << 12);
      if (KIND == 0) { f32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); asm volatile("" :: "v"(v)); }
      else if (KIND == 1) { f32x2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); asm volatile("" :: "v"(v)); }
      else { f32x2 v; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a)); asm volatile("" :: "v"(v)); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();   // every wave of the workgroup is done (the oldest wave wins the arbitration and would finish early on its own)
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (acc == 12345.678f) out[63] = 1;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  unsigned long long* out; unsigned long long h[512];
  hipMalloc((void**)&out, 4096);
  const int iters = 4000;
  for (int kind = 0; kind < 3; ++kind)
    for (int waves : {4, 8, 16}) {
      auto k = kind == 0 ? probe<0> : (kind == 1 ? probe<1> : probe<2>);
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, out, iters);
      hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 65536, 0, out, iters);
      hipDeviceSynchronize();
      hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
      const double bytes = (double)iters * 16 * waves * 64 * (kind == 0 ? 16 : 8);
      printf("%s waves %2d: %6.1f B per s_memtime tick per CU, %7.1f B per 10-ns tick (s_memtime / realtime ticks = %.2f)\n",
             kind == 0 ? "ds_read_b128      " : (kind == 1 ? "ds_read_b64       " : "ds_read_b64_tr_b16"), waves, bytes / (double)h[0], bytes / (double)h[1],
             (double)h[0] / (double)h[1]);
    }
  return 0;
}
