// Hardware probe: range checking of raw buffer loads (stride 0) on gfx950 -- which 16-B accesses return zeros for a given num_records?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, float* o, int nbytes, int flags) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, flags);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 0);
  o[threadIdx.x * 4 + 0] = __builtin_bit_cast(float, v.x); o[threadIdx.x * 4 + 1] = __builtin_bit_cast(float, v.y);
  o[threadIdx.x * 4 + 2] = __builtin_bit_cast(float, v.z); o[threadIdx.x * 4 + 3] = __builtin_bit_cast(float, v.w);
}
int main() {
  float *p, *o, h[256], r[256];
  hipMalloc((void**)&p, 1024); hipMalloc((void**)&o, 1024);
  for (int i = 0; i < 256; ++i) h[i] = 1.f + i;
  hipMemcpy(p, h, 1024, hipMemcpyHostToDevice);
  for (int flags : {0x00020000, 0x00027000}) for (int nbytes : {64, 72, 80}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o, nbytes, flags);
    hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    printf("flags %#x num_records %d:", flags, nbytes);
    for (int i = 0; i < 32; ++i) printf(" %g", r[i]);
    printf("\n");
  }
  return 0;
}
