// Hardware probe: how many bytes per clock can ONE CU pull through (a) the LDS-DMA path (buffer_load_dwordx4 ... lds, 1 KiB per wave
// instruction) and (b) ordinary buffer_load_dwordx4 into VGPRs, as a function of the access shape of a GEMM operand tile
// (8 rows x 128 B = full cache lines, BK = 64;  16 rows x 64 B = half lines, BK = 32), the number of issuing waves and the residency of the
// source (a 2 MiB region per workgroup re-read every pass = L2 / MALL hits;  a fresh 384 MiB stream = HBM).  One workgroup per CU, every
// wave keeps DEPTH pieces in flight (counted vmcnt).  Prints bytes / clock / CU (s_memtime ticks at 100 MHz are converted with the measured
// kernel time; the clock is taken as 2.1 GHz for the per-clock figure) and GB/s per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_rsrc(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
}
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned dst, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// MODE 0: LDS-DMA, 1: buffer_load to VGPR (results xor-folded so the loads stay).  HALF: 0 = 8 rows x 128 B per piece, 1 = 16 rows x 64 B.
template <int MODE, int HALF, int DEPTH>
__global__ void probe(const char* src, long long region_bytes, long long wg_stride, int row_bytes, int pieces_per_wave, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const char* base = src + (long long)blockIdx.x * wg_stride;
  const i32x4 rs = make_rsrc(base);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  // lane -> (row, 16-B chunk) inside a piece
  const int row = HALF ? (lane >> 2) : (lane >> 3);
  const int chunk = HALF ? (lane & 3) : (lane & 7);
  const unsigned voff = (unsigned)(row * row_bytes + chunk * 16);
  const int rows_per_piece = HALF ? 16 : 8;
  u32x4 acc = {0, 0, 0, 0};
  // the K-tile stream of 256 x 128 output tiles: 384 operand rows per K-tile, K-tiles walk along the row, then the next 384 rows of the region.
  // pieces_per_wave is used as the number of TILES each workgroup walks; all address arithmetic is incremental (no divisions in the loop).
  const int colB = HALF ? 64 : 128, ppi = 384 / rows_per_piece, nkt = row_bytes / colB;
  const unsigned region_tiles = (unsigned)(region_bytes / ((long long)384 * row_bytes));
  int issued = 0;
  unsigned rt = 0;
  for (int tile = 0; tile < pieces_per_wave; ++tile) {
    const unsigned tile_off = rt * 384u * (unsigned)row_bytes;
    if (++rt == region_tiles) rt = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      for (int rg = wave; rg < ppi; rg += nw) {
        const unsigned off = tile_off + (unsigned)(rg * rows_per_piece) * (unsigned)row_bytes + (unsigned)(kt * colB);
        if (MODE == 0) {
          dma16(rs, lds0 + (unsigned)(((issued % DEPTH) * nw + wave) * 1024), voff, off);
        } else {
          u32x4 v;
          asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(off) : "memory");
          asm volatile("" : "+v"(v));
          acc = v;
        }
        if (issued >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(DEPTH - 1) : "memory");
        ++issued;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
  if (MODE == 0 && smem[lane] == 0x7f && smem[lane + 64] == 0x7e) sink[1] = 1;
}

int main() {
  char* src;
  unsigned* sink;
  const size_t total = 1024ull << 20;
  hipMalloc((void**)&src, total + (1 << 20));
  hipMemset(src, 1, total);
  hipMalloc((void**)&sink, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](auto kern, const char* tag, int waves, int row_bytes, long long region, long long wg_stride, int pieces) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 128 * 1024, 0, src, region, wg_stride, row_bytes, pieces, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double bytes_cu = (double)pieces * (row_bytes / 128) * 384.0 * 128.0;   // pieces = tiles per workgroup; a tile = 384 rows x row_bytes
    const double gbs = bytes_cu / (best * 1e-3) / 1e9;
    printf("%-44s waves %d  %7.1f GB/s per CU  %5.1f B/clk/CU @2.1GHz  (chip %.2f TB/s, %.3f ms)\n", tag, waves, gbs, gbs / 2.1, gbs * 256 / 1e3, best);
  };
  // shared: every workgroup reads the SAME 1152-row panel (weights-like: L2 hits);  own: each workgroup its own 384-row panel, 147 MB in all
  // (re-read: memory-side cache hits);  stream: each workgroup its own 3456 rows once (885 MB: HBM)
  const int rowb = 1536;
  for (int waves : {4, 8}) {
    const int pw = 32;   // tiles (384 rows x 1536 B = 576 KiB each) per workgroup
    run(probe<0, 0, 8>, "DMA  full-line  shared (L2)   depth 8", waves, rowb, 1152ll * rowb, 0, pw);
    run(probe<0, 1, 8>, "DMA  half-line  shared (L2)   depth 8", waves, rowb, 1152ll * rowb, 0, pw);
    run(probe<0, 0, 16>, "DMA  full-line  shared (L2)   depth 16", waves, rowb, 1152ll * rowb, 0, pw);
    run(probe<1, 0, 8>, "VGPR full-line  shared (L2)   depth 8", waves, rowb, 1152ll * rowb, 0, pw);
    run(probe<1, 1, 8>, "VGPR half-line  shared (L2)   depth 8", waves, rowb, 1152ll * rowb, 0, pw);
    run(probe<1, 0, 16>, "VGPR full-line  shared (L2)   depth 16", waves, rowb, 1152ll * rowb, 0, pw);
    run(probe<0, 0, 8>, "DMA  full-line  own 576KB (MALL) depth 8", waves, rowb, 384ll * rowb, 384ll * rowb, pw);
    run(probe<0, 1, 8>, "DMA  half-line  own 576KB (MALL) depth 8", waves, rowb, 384ll * rowb, 384ll * rowb, pw);
    run(probe<0, 0, 16>, "DMA  full-line  own 576KB (MALL) depth 16", waves, rowb, 384ll * rowb, 384ll * rowb, pw);
    run(probe<1, 0, 8>, "VGPR full-line  own 576KB (MALL) depth 8", waves, rowb, 384ll * rowb, 384ll * rowb, pw);
    run(probe<1, 0, 16>, "VGPR full-line  own 576KB (MALL) depth 16", waves, rowb, 384ll * rowb, 384ll * rowb, pw);
    run(probe<0, 0, 8>, "DMA  full-line  stream (HBM)  depth 8", waves, rowb, 2304ll * rowb, 2304ll * rowb, pw / 2);
    run(probe<1, 0, 8>, "VGPR full-line  stream (HBM)  depth 8", waves, rowb, 2304ll * rowb, 2304ll * rowb, pw / 2);
  }
  return 0;
}
