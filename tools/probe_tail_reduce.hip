// Measurement behind DESIGN.md section 5, round 5, item 7 (weight gradients: "last-arriving K slice of a tile sums its siblings").
// The split-K weight-gradient GEMM leaves S fp32 partial slices of dW[in][out]; today a separate launch over the whole chip sums them in a fixed
// order (kernel A).  The in-kernel form has ONE workgroup per 256 x 256 output tile -- the last slice to finish -- read all S slices of its tile and
// write the sum while every other CU has already left the kernel (kernel B: exactly that access pattern, 512 threads per tile, float4, fixed order).
// Shapes: the four weight gradients of a ViT-B/16 block at batch 256 with the slice counts the engine uses (bench.py gemm_shapes).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_tail_reduce tools/probe_tail_reduce.hip && tools/probe_tail_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void reduce_all(const float4* __restrict__ part, int slices, long stride4, long n4, float4* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 a = part[i];
  for (int s = 1; s < slices; ++s) {
    const float4 v = part[s * stride4 + i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  out[i] = a;
}

// one workgroup per 256 x 256 tile of the [rows][cols] matrix: 64 float4 per tile row
__global__ __launch_bounds__(512) void reduce_tile(const float4* __restrict__ part, int slices, long stride4, int cols4, int tiles_n, float4* __restrict__ out) {
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  for (int e = threadIdx.x; e < 256 * 64; e += 512) {
    const int r = e >> 6, c = e & 63;
    const long i = (long)(tm * 256 + r) * cols4 + tn * 64 + c;
    float4 a = part[i];
    for (int s = 1; s < slices; ++s) {
      const float4 v = part[s * stride4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    out[i] = a;
  }
}

int main() {
  struct Shape { const char* name; int rows, cols, slices; };
  const Shape shapes[] = {{"out-proj 768 x 768, 28 slices", 768, 768, 28}, {"qkv 768 x 2304, 9 slices", 768, 2304, 9},
                          {"fc1 768 x 3072, 7 slices", 768, 3072, 7}, {"fc2 3072 x 768, 7 slices", 3072, 768, 7}};
  float *part, *out;
  const size_t maxn = (size_t)768 * 3072;
  hipMalloc(&part, maxn * 28 * 4);
  hipMalloc(&out, maxn * 4);
  hipMemset(part, 0, maxn * 28 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  char* flush; hipMalloc(&flush, (size_t)1 << 30);
  for (const Shape& sh : shapes) {
    const long n = (long)sh.rows * sh.cols, n4 = n / 4;
    const int tiles_m = sh.rows / 256, tiles_n = sh.cols / 256;
    float ta = 0.f, tb = 0.f;
    const int reps = 20;
    for (int rep = 0; rep < reps + 2; ++rep) {
      float ms;
      // the partials were just written by the GEMM: leave them wherever a 1-GB sweep leaves them (not in L2), as after a 200-us GEMM of other traffic
      hipMemsetAsync(flush, rep, (size_t)1 << 30, 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(reduce_all, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float4*)part, sh.slices, n4, n4, (float4*)out);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2) ta += ms;
      hipMemsetAsync(flush, rep, (size_t)1 << 30, 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(reduce_tile, dim3(tiles_m * tiles_n), dim3(512), 0, 0, (const float4*)part, sh.slices, n4, sh.cols / 4, tiles_n, (float4*)out);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2) tb += ms;
    }
    printf("%-34s %3d tiles: whole-chip reduction launch %6.1f us | one workgroup per tile (the last-arriver form) %6.1f us\n", sh.name,
           tiles_m * tiles_n, ta / reps * 1e3, tb / reps * 1e3);
  }
  return 0;
}
