// Hardware probe: how many 512-thread workgroups with 56 KiB of LDS does one gfx950 CU hold as a function of the kernel's VGPR count?
// (The attention kernels are sized for two; the forward at 127 VGPRs got one -- tools/probe_attn.)  Each kernel touches VGPR N-1 so that its
// descriptor asks for N registers, spins ~10 us, and records its start time; the number of workgroups started before the first one ends is
// the residency.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_occupancy.hip -o tools/probe_occupancy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define PROBE_KERNEL(N, REG)                                                                              \
  __global__ __launch_bounds__(512) void probe_##N(unsigned long long* out) {                             \
    extern __shared__ char smem[];                                                                        \
    asm volatile("v_mov_b32 " REG ", 0" ::: REG);                                                         \
    const unsigned long long t0 = wall_clock64();                                                         \
    if (threadIdx.x == 0) smem[0] = 1;                                                                    \
    while (wall_clock64() - t0 < 1000) { }                                                                \
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t0; out[blockIdx.x * 2 + 1] = wall_clock64(); }         \
  }
PROBE_KERNEL(96, "v95")
PROBE_KERNEL(104, "v103")
PROBE_KERNEL(112, "v111")
PROBE_KERNEL(120, "v119")
PROBE_KERNEL(124, "v123")
PROBE_KERNEL(127, "v126")
PROBE_KERNEL(128, "v127")
PROBE_KERNEL(136, "v135")

template <typename K>
int run(K kern, int nv, int lds, unsigned long long* dev, int nwg) {
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  std::vector<unsigned long long> st(nwg * 2);
  for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, 0, dev); CK(hipDeviceSynchronize()); }
  CK(hipMemcpy(st.data(), dev, st.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long first_end = ~0ull;
  for (int w = 0; w < nwg; ++w) first_end = std::min(first_end, st[w * 2 + 1]);
  int resident = 0;
  for (int w = 0; w < nwg; ++w) resident += st[w * 2] < first_end;
  printf("VGPRs %3d  LDS %6d B  threads 512: %4d workgroups resident at once = %.2f per CU\n", nv, lds, resident, resident / 256.0);
  return 0;
}

int main() {
  const int nwg = 2048;
  unsigned long long* dev;
  CK(hipMalloc(&dev, nwg * 16));
  for (int lds : {57344, 1024}) {
    run(probe_96, 96, lds, dev, nwg); run(probe_104, 104, lds, dev, nwg); run(probe_112, 112, lds, dev, nwg); run(probe_120, 120, lds, dev, nwg);
    run(probe_124, 124, lds, dev, nwg); run(probe_127, 127, lds, dev, nwg); run(probe_128, 128, lds, dev, nwg); run(probe_136, 136, lds, dev, nwg);
  }
  return 0;
}
