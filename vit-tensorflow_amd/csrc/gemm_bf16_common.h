// Pieces shared by the three bf16 MFMA GEMM translation units (gemm_bf16.hip: variant dispatch + the lockstep NT kernel,
// gemm_bf16_pipe.hip: the software-pipelined persistent NT kernel, gemm_bf16_tn.hip: the weight-gradient kernel).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

constexpr int BK = 64;

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int V> using ic = std::integral_constant<int, V>;

// logical tile index -> (tile_m, tile_n).  Wide outputs (>= 8 column tiles) are walked in bands of 4 row tiles, column-major inside a
// band, so the ~32 consecutive tiles an XCD works on at any time form a 4 x 8 block: 12 operand panels in flight instead of 2.7 + 12
// (PMC: the row-major order re-fetched the A panel of fc1 5.5x and of qkv 3.7x through the XCD's 4 MiB L2; 8192^3 +12 %).
__device__ __forceinline__ void decode_tile_fwd(int L, int tiles_m, int tiles_n, int gm, int& tm, int& tn);
// gm < 0: the same walk with the row tiles taken from the LAST to the first.  For a GEMM whose A operand was just written, front to back,
// by the previous kernel and is larger than the 256 MB memory-side cache (act for fc2, d(hpre) for the fc1 dgrad: 310 MB): the cache holds
// the most recently written rows, and a reader that starts at row 0 misses, allocates, and evicts exactly the rows it needs next; starting
// at the end it hits on everything that is still there.
__device__ __forceinline__ void decode_tile(int L, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
  if (gm < 0) { decode_tile_fwd(L, tiles_m, tiles_n, -gm, tm, tn); tm = tiles_m - 1 - tm; return; }
  decode_tile_fwd(L, tiles_m, tiles_n, gm, tm, tn);
}
__device__ __forceinline__ void decode_tile_fwd(int L, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
  if (gm == 1) { tm = L / tiles_n; tn = L - tm * tiles_n; return; }
  const int group = gm * tiles_n;
  const int gid = L / group, first = gid * gm;
  const int gsz = min(tiles_m - first, gm);
  const int w = L - gid * group;
  tn = w / gsz;
  tm = first + (w - tn * gsz);
}

}  // namespace

// gemm_bf16_pipe.hip: the pipelined persistent NT kernel behind the variant numbers of launch_gemm_bf16 (9, 12, 13, 14, 15)
void launch_gemm_bf16_pipe(int variant, int mode, const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s);
// gemm_bf16.hip: 1 = collectives share the CUs (a gradient-ready callback is registered): no persistent grids
int gemm_bf16_shared_gpu();
// gemm_bf16.hip: the persistent lockstep kernel (variants 6 / 7), the pipelined kernel's fall-back for operands beyond 2 GiB
void launch_gemm_bf16_persistent_lockstep(int bm, int mode, const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s);
