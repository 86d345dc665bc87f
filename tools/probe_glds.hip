// Hardware probe: which counter(s) cover global_load_lds (LDS-DMA)?  One wave issues N 1-KiB DMA pieces from cold memory and
// timestamps (s_memtime) after  s_waitcnt lgkmcnt(0)  and after  s_waitcnt vmcnt(0).
// Also: DMA landing rate per CU with 1..8 waves issuing (KiB pieces back to back), to size the GEMM prefetch ring.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int N>
__global__ void probe(const char* src, long long* out, int stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* p = src + (size_t)blockIdx.x * (64u << 20) / 256 + (size_t)wave * N * stride + lane * 16;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; ++i)
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(p + (size_t)i * stride), (lds_void_t*)(smem + (wave * N + i) * 1024), 16, 0, 0);
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t2 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t3 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x == 0) {
    out[wave * 4 + 0] = t1 - t0; out[wave * 4 + 1] = t2 - t0; out[wave * 4 + 2] = t3 - t0; out[wave * 4 + 3] = smem[lane];
  }
}

int main() {
  char* src; long long* out;
  hipMalloc((void**)&src, 512u << 20);
  hipMemset(src, 1, 512u << 20);
  hipMalloc((void**)&out, 4096);
  std::vector<long long> h(64);
  auto run = [&](auto kern, int waves, int blocks, const char* tag, int n) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 128 * 1024, 0, src + (size_t)rep * (128u << 20), out, 4096);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), out, 512, hipMemcpyDeviceToHost);
    printf("%s: waves %d blocks %d pieces/wave %d : issue %lld  after lgkmcnt(0) %lld  after vmcnt(0) %lld cycles (100 MHz ref clock ticks x?)\n",
           tag, waves, blocks, n, h[0], h[1], h[2]);
  };
  run(probe<1>, 1, 1, "single piece", 1);
  run(probe<8>, 1, 1, "8 pieces 1 wave", 8);
  run(probe<8>, 8, 1, "64 KB / 8 waves, 1 CU", 8);
  run(probe<16>, 8, 1, "128 KB / 8 waves, 1 CU", 16);
  run(probe<8>, 8, 256, "64 KB / 8 waves, 256 CUs", 8);
  run(probe<16>, 8, 256, "128 KB / 8 waves, 256 CUs", 16);
  run(probe<8>, 4, 256, "32 KB / 4 waves, 256 CUs", 8);
  return 0;
}
