// Hardware probe: prints the lane/element mapping of ds_read_b64_tr_b16 and checks the MFMA fragment
// layouts the kernels assume (16x16x32 / 32x32x16 bf16, asymmetric operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void tr_probe(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = t[i];
}
// D = A(16x32) * B(32x16) with A[i][k] = i + 0.25k (k<4 only nonzero pattern avoided: full), B[k][j] = (k==j?1:0)+ 0.5*(k==j+16)
__global__ void mfma16_probe(const float* A, const float* B, float* D) {   // A [16][32], B [32][16] row-major fp32
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    int k = (l >> 4) * 8 + e;
    a[e] = (__bf16)A[(l & 15) * 32 + k];
    b[e] = (__bf16)B[k * 16 + (l & 15)];
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void mfma32_probe(const float* A, const float* B, float* D) {   // A [32][16], B [16][32]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    int k = (l >> 5) * 8 + e;
    a[e] = (__bf16)A[(l & 31) * 16 + k];
    b[e] = (__bf16)B[k * 32 + (l & 31)];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
  short* d; hipMalloc(&d, 512);
  tr_probe<<<1, 64>>>(d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("ds_read_b64_tr_b16 with lane l addressing elements [4l,4l+4): lane -> 4 element indices\n");
  for (int l = 0; l < 64; ++l) printf("L%02d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  // expected by the guide: lane l elem j = (l&15) + 16 j + 64 (l>>4)
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) ok &= (h[l * 4 + j] == (l & 15) + 16 * j + 64 * (l >> 4));
  printf("tr_b16 matches guide formula: %s\n", ok ? "YES" : "NO");
  for (int shape = 0; shape < 2; ++shape) {
    int M = shape ? 32 : 16, K = shape ? 16 : 32, N = M;
    std::vector<float> A(M * K), B(K * N), Dh(M * N), R(M * N, 0.f);
    for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) A[i * K + k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < K; ++k) for (int j = 0; j < N; ++j) B[k * N + j] = (float)((k * 5 + j * 2) % 7 - 3);
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { float s = 0; for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * N + j]; R[i * N + j] = s; }
    float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, Dh.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    if (shape) mfma32_probe<<<1, 64>>>(dA, dB, dD); else mfma16_probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(Dh.data(), dD, Dh.size() * 4, hipMemcpyDeviceToHost);
    float me = 0; for (int i = 0; i < M * N; ++i) me = fmaxf(me, fabsf(Dh[i] - R[i]));
    printf("mfma %s layout check: max err %g -> %s\n", shape ? "32x32x16" : "16x16x32", me, me == 0.f ? "OK" : "MISMATCH");
  }
  return 0;
}
