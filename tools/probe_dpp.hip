#include "../vit-tensorflow_amd/csrc/common.h"
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, float* out) {
  float v = in[threadIdx.x];
  out[threadIdx.x] = wave_sum(v);
  out[64 + threadIdx.x] = wave_max(v);
}
int main() {
  float h[64], *d, *o, r[128];
  double s = 0; float m = -1e30f;
  for (int i = 0; i < 64; ++i) { h[i] = sinf(i * 1.7f) * 3.f + i * 0.01f; s += h[i]; m = fmaxf(m, h[i]); }
  hipMalloc((void**)&d, 256); hipMalloc((void**)&o, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) { if (fabs(r[i] - s) > 1e-4) bad++; if (r[64 + i] != m) bad++; }
  printf("sum %f (ref %f) max %f (ref %f) bad %d\n", r[0], s, r[64], m, bad);
  return bad != 0;
}
