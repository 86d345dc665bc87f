// Index / masking / reconstruction-loss kernels of the masked-image-modelling wrappers that call
// encoder.transformer(tokens) on a subset of the patches (mae.py:47-92, simmim.py:86-130).  They replace the reference's host-side
// `.numpy()[batch_range, indices]` fancy indexing (mae.py:62,65; simmim.py:119,125), scatter_numpy (simmim.py:9-66,109) and
// tf.where (simmim.py:113) by device gathers, so the token stream never leaves HBM -- and, unlike the reference's round trip through
// numpy, stays differentiable.  All of it is HBM-bound row copying: one thread per 16 B (4 B when the row width is not a multiple
// of 4), consecutive lanes on consecutive addresses; every reduction runs in a fixed order (no atomics, bit-reproducible).
//
// Index convention: idx is int32 [b, ldi]; a "position range" [j0, j1) selects the columns of idx an op looks at; inv is the
// inverse map int32 [b, n] (inv[b, idx[b, j]] = j for j in the range the inverse was built for, -1 elsewhere).
#include "kernels.h"

namespace {

template <int V> struct vec_t;
template <> struct vec_t<4> { typedef float4 type; };
template <> struct vec_t<1> { typedef float type; };
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
template <typename T> __device__ __forceinline__ T vzero();
template <> __device__ __forceinline__ float4 vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float vzero<float>() { return 0.f; }

__global__ void index_inverse_kernel(const int32_t* __restrict__ idx, int64_t ldi, int b, int j0, int j1, int n, int32_t* __restrict__ inv) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = j1 - j0;
  if (e >= (int64_t)b * k) return;
  const int bi = (int)(e / k), j = j0 + (int)(e - (int64_t)bi * k);
  const int t = idx[bi * ldi + j];
  if (t >= 0 && t < n) inv[(int64_t)bi * n + t] = j;
}

// out[bi, j, :] = src[bi * src_bstride + idx[bi, j0 + j] * dv + :]      (src_bstride = 0: one table shared by the batch)
template <int V>
__global__ void gather_rows_kernel(const typename vec_t<V>::type* __restrict__ src, int64_t src_bstride, const int32_t* __restrict__ idx,
                                   int64_t ldi, int j0, int b, int k, int dv, typename vec_t<V>::type* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)b * k * dv) return;
  const int64_t row = e / dv;
  const int c = (int)(e - row * dv);
  const int bi = (int)(row / k), j = (int)(row - (int64_t)bi * k);
  const int t = idx[bi * ldi + j0 + j];
  out[e] = src[bi * src_bstride + (int64_t)t * dv + c];
}

// dst[bi, t, :] = inv[bi, t] in [j0, j1) ? src[bi, inv[bi, t] - j0, :] : 0        (the VJP of the gather above when src is per-image)
template <int V>
__global__ void scatter_rows_kernel(const typename vec_t<V>::type* __restrict__ src, int k_src, const int32_t* __restrict__ inv, int j0, int j1,
                                    int b, int n, int dv, typename vec_t<V>::type* __restrict__ dst) {
  typedef typename vec_t<V>::type T;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)b * n * dv) return;
  const int64_t row = e / dv;
  const int c = (int)(e - row * dv);
  const int bi = (int)(row / n);
  const int j = inv[row];
  dst[e] = (j >= j0 && j < j1) ? src[((int64_t)bi * k_src + (j - j0)) * dv + c] : vzero<T>();
}

// dtab[t, :] (= | +=) sum_bi (inv[bi, t] in [j0, j1) ? src[bi, inv[bi, t] - j0, :] : 0)      (VJP of a shared-table gather)
template <int V>
__global__ void table_grad_kernel(const typename vec_t<V>::type* __restrict__ src, int k_src, const int32_t* __restrict__ inv, int j0, int j1,
                                  int b, int n, int dv, int accumulate, typename vec_t<V>::type* __restrict__ dtab) {
  typedef typename vec_t<V>::type T;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * dv) return;
  const int t = (int)(e / dv), c = (int)(e - (int64_t)t * dv);
  T a = accumulate ? dtab[e] : vzero<T>();
  for (int bi = 0; bi < b; ++bi) {
    const int j = inv[(int64_t)bi * n + t];
    if (j >= j0 && j < j1) a = vadd(a, src[((int64_t)bi * k_src + (j - j0)) * dv + c]);
  }
  dtab[e] = a;
}

// MAE decoder input (mae.py:72-82): out[bi, j] = (j < nm ? mask_token : proj[bi, j - nm]) + dpos[idx[bi, j]]
template <int V>
__global__ void mae_assemble_kernel(const typename vec_t<V>::type* __restrict__ proj, const typename vec_t<V>::type* __restrict__ mask_token,
                                    const typename vec_t<V>::type* __restrict__ dpos, const int32_t* __restrict__ idx, int b, int np, int nm,
                                    int dv, typename vec_t<V>::type* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)b * np * dv) return;
  const int64_t row = e / dv;
  const int c = (int)(e - row * dv);
  const int bi = (int)(row / np), j = (int)(row - (int64_t)bi * np);
  const int t = idx[row];
  const auto base = j < nm ? mask_token[c] : proj[((int64_t)bi * (np - nm) + (j - nm)) * dv + c];
  out[e] = vadd(base, dpos[(int64_t)t * dv + c]);
}

// partial[bi, :] = sum over the selected rows t of x[bi, t, :]; selected = inv ? inv[bi, t] >= 0 : j0 <= t < j1
template <int V>
__global__ void select_rowsum_kernel(const typename vec_t<V>::type* __restrict__ x, const int32_t* __restrict__ inv, int j0, int j1, int n, int dv,
                                     typename vec_t<V>::type* __restrict__ partial) {
  typedef typename vec_t<V>::type T;
  const int bi = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= dv) return;
  T a = vzero<T>();
  for (int t = 0; t < n; ++t) {
    const bool sel = inv ? inv[(int64_t)bi * n + t] >= 0 : (t >= j0 && t < j1);
    if (sel) a = vadd(a, x[((int64_t)bi * n + t) * dv + c]);
  }
  partial[(int64_t)bi * dv + c] = a;
}

// SimMIM token replacement (simmim.py:102-113): x[bi, t] = inv[bi, t] >= 0 ? mask_token + pos[t] : x[bi, t]
template <int V>
__global__ void simmim_select_kernel(typename vec_t<V>::type* __restrict__ x, const int32_t* __restrict__ inv,
                                     const typename vec_t<V>::type* __restrict__ mask_token, const typename vec_t<V>::type* __restrict__ pos, int b,
                                     int n, int dv) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)b * n * dv) return;
  const int64_t row = e / dv;
  const int c = (int)(e - row * dv);
  const int t = (int)(row % n);
  if (inv[row] >= 0) x[e] = vadd(mask_token[c], pos[(int64_t)t * dv + c]);
}

template <int V>
__global__ void zero_selected_rows_kernel(typename vec_t<V>::type* __restrict__ x, const int32_t* __restrict__ inv, int64_t rows, int dv) {
  typedef typename vec_t<V>::type T;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * dv) return;
  if (inv[e / dv] >= 0) x[e] = vzero<T>();
}

// reconstruction losses: kind 0 = squared (mae.py:90), kind 1 = absolute (simmim.py:128).  target may be null (treated as 0).
// dpred = scale * f'(pred - target); partial[block] = scale * sum f(pred - target) over the block's elements.
constexpr int LOSS_THREADS = 256, LOSS_PER_THREAD = 8;
__global__ __launch_bounds__(LOSS_THREADS) void recon_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t count,
                                                                  int kind, float scale, float* __restrict__ dpred, float* __restrict__ partial) {
  __shared__ float red[LOSS_THREADS / 64];
  const int64_t base = (int64_t)blockIdx.x * (LOSS_THREADS * LOSS_PER_THREAD);
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < LOSS_PER_THREAD; ++i) {
    const int64_t e = base + i * LOSS_THREADS + threadIdx.x;
    if (e < count) {
      const float diff = pred[e] - (target ? target[e] : 0.f);
      if (kind == 0) { a += diff * diff; dpred[e] = 2.f * diff * scale; }
      else if (kind == 2) { a += diff; dpred[e] = scale; }   // the MPP loss as written (mpp.py:125): a linear functional of the logits
      else { a += fabsf(diff); dpred[e] = (float)((diff > 0.f) - (diff < 0.f)) * scale; }
    }
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) s += red[w];
    partial[blockIdx.x] = s * scale;
  }
}
__global__ __launch_bounds__(64) void final_sum_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) a += partial[i];   // fixed assignment of partials to lanes, fixed DPP tree
  a = wave_sum(a);
  if (threadIdx.x == 0) *out = a;
}

// dst[b][row0 + idx[b][j]][:] = src[b][j][:] (rows not named by idx are left as they are: the caller zero-fills)
__global__ void scatter_by_index_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int b, int k, int d, float* __restrict__ dst,
                                        int64_t dst_batch_stride, int row0) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)b * k * d) return;
  const int c = (int)(e % d);
  const int64_t r = e / d;
  const int bi = (int)(r / k);
  dst[(int64_t)bi * dst_batch_stride + (int64_t)(row0 + idx[r]) * d + c] = src[e];
}
// MPPLoss target labels of the masked patches (mpp.py:104-123): mean colour of the patch per channel (after the optional un-normalisation
// and the clamp to [0, max_pixel_val]), Bucketize against arange(bin, mpv, bin) (= number of boundaries <= value), label = sum_c (2^bits)^c bucket_c
__global__ void mpp_labels_kernel(const float* __restrict__ img, int H, int W, int C, int p, int bits, float mpv, int has_norm, float4 mean4, float4 std4,
                                  const int32_t* __restrict__ idx, int b, int k, int32_t* __restrict__ labels) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= b * k) return;
  const int bi = r / k, wp = W / p, t = idx[r], py = t / wp, px = t - py * wp;
  const float mean[4] = {mean4.x, mean4.y, mean4.z, mean4.w}, sd[4] = {std4.x, std4.y, std4.z, std4.w};
  const float bin = mpv / (float)(1 << bits);
  const int nbound = (int)ceilf((mpv - bin) / bin - 1e-6f);   // len(np.arange(bin, mpv, bin))
  int label = 0, mul = 1;
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
    for (int y = 0; y < p; ++y)
      for (int x = 0; x < p; ++x) {
        float v = img[(((int64_t)bi * H + py * p + y) * W + px * p + x) * C + c];
        if (has_norm) v = v * sd[c] + mean[c];
        acc += fminf(fmaxf(v, 0.f), mpv);
      }
    const float avg = acc / (float)(p * p);
    int bucket = 0;
    for (int j = 1; j <= nbound; ++j) bucket += (bin * (float)j <= avg) ? 1 : 0;
    label += mul * bucket;
    mul <<= bits;
  }
  labels[r] = label;
}

inline unsigned blocks_for(int64_t total) { return (unsigned)ceil_div(total, 256); }   // one element per thread, no grid-stride loop
inline bool vec_ok(int d, std::initializer_list<const void*> ptrs) {
  if (d % 4) return false;
  for (const void* p : ptrs) if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

}  // namespace

#define MIM_DISPATCH(KERNEL, GRID, BLOCK, VEC, ...)                                                                     \
  do {                                                                                                                  \
    if (VEC) hipLaunchKernelGGL(KERNEL<4>, GRID, BLOCK, 0, s, __VA_ARGS__);                                             \
    else hipLaunchKernelGGL(KERNEL<1>, GRID, BLOCK, 0, s, __VA_ARGS__);                                                 \
  } while (0)
#define F4(p) ((const float4*)(p))
#define F4M(p) ((float4*)(p))

void launch_index_inverse(const int32_t* idx, int64_t ldi, int b, int j0, int j1, int n, int32_t* inv, hipStream_t s) {
  (void)hipMemsetAsync(inv, 0xFF, (size_t)b * n * 4, s);   // -1 everywhere
  const int64_t total = (int64_t)b * (j1 - j0);
  if (total > 0) hipLaunchKernelGGL(index_inverse_kernel, dim3(blocks_for(total)), dim3(256), 0, s, idx, ldi, b, j0, j1, n, inv);
}

void launch_gather_rows(const float* src, int64_t src_batch_stride, const int32_t* idx, int64_t ldi, int j0, int b, int k, int d, float* out,
                        hipStream_t s) {
  if ((int64_t)b * k * d == 0) return;
  if (vec_ok(d, {src, out}) && src_batch_stride % 4 == 0)
    hipLaunchKernelGGL(gather_rows_kernel<4>, dim3(blocks_for((int64_t)b * k * (d / 4))), dim3(256), 0, s, F4(src), src_batch_stride / 4, idx, ldi, j0, b,
                       k, d / 4, F4M(out));
  else
    hipLaunchKernelGGL(gather_rows_kernel<1>, dim3(blocks_for((int64_t)b * k * d)), dim3(256), 0, s, src, src_batch_stride, idx, ldi, j0, b, k, d, out);
}

void launch_scatter_rows(const float* src, int k_src, const int32_t* inv, int j0, int j1, int b, int n, int d, float* dst, hipStream_t s) {
  if ((int64_t)b * n * d == 0) return;
  if (vec_ok(d, {src, dst}))
    hipLaunchKernelGGL(scatter_rows_kernel<4>, dim3(blocks_for((int64_t)b * n * (d / 4))), dim3(256), 0, s, F4(src), k_src, inv, j0, j1, b, n, d / 4,
                       F4M(dst));
  else
    hipLaunchKernelGGL(scatter_rows_kernel<1>, dim3(blocks_for((int64_t)b * n * d)), dim3(256), 0, s, src, k_src, inv, j0, j1, b, n, d, dst);
}

void launch_table_grad(const float* src, int k_src, const int32_t* inv, int j0, int j1, int b, int n, int d, int accumulate, float* dtab,
                       hipStream_t s) {
  if ((int64_t)n * d == 0) return;
  if (vec_ok(d, {src, dtab}))
    hipLaunchKernelGGL(table_grad_kernel<4>, dim3((unsigned)ceil_div((int64_t)n * (d / 4), 64)), dim3(64), 0, s, F4(src), k_src, inv, j0, j1, b, n,
                       d / 4, accumulate, F4M(dtab));
  else
    hipLaunchKernelGGL(table_grad_kernel<1>, dim3((unsigned)ceil_div((int64_t)n * d, 64)), dim3(64), 0, s, src, k_src, inv, j0, j1, b, n, d, accumulate,
                       dtab);
}

void launch_mae_assemble(const float* proj, const float* mask_token, const float* dpos, const int32_t* idx, int b, int np, int nm, int d, float* out,
                         hipStream_t s) {
  if ((int64_t)b * np * d == 0) return;
  if (vec_ok(d, {proj, mask_token, dpos, out}))
    hipLaunchKernelGGL(mae_assemble_kernel<4>, dim3(blocks_for((int64_t)b * np * (d / 4))), dim3(256), 0, s, F4(proj), F4(mask_token), F4(dpos), idx, b, np,
                       nm, d / 4, F4M(out));
  else
    hipLaunchKernelGGL(mae_assemble_kernel<1>, dim3(blocks_for((int64_t)b * np * d)), dim3(256), 0, s, proj, mask_token, dpos, idx, b, np, nm, d, out);
}

// out[:] = sum over images and selected rows of x[bi, t, :]   (partial_ws: b * d floats)
void launch_select_rowsum(const float* x, const int32_t* inv_or_null, int j0, int j1, int b, int n, int d, float* partial_ws, float* out,
                          hipStream_t s) {
  if (vec_ok(d, {x, partial_ws}))
    hipLaunchKernelGGL(select_rowsum_kernel<4>, dim3((unsigned)ceil_div(d / 4, 64), (unsigned)b), dim3(64), 0, s, F4(x), inv_or_null, j0, j1, n, d / 4,
                       F4M(partial_ws));
  else
    hipLaunchKernelGGL(select_rowsum_kernel<1>, dim3((unsigned)ceil_div(d, 64), (unsigned)b), dim3(64), 0, s, x, inv_or_null, j0, j1, n, d, partial_ws);
  launch_sum_rows(partial_ws, b, d, out, s);
}

void launch_simmim_select(float* x, const int32_t* inv, const float* mask_token, const float* pos, int b, int n, int d, hipStream_t s) {
  if ((int64_t)b * n * d == 0) return;
  if (vec_ok(d, {x, mask_token, pos}))
    hipLaunchKernelGGL(simmim_select_kernel<4>, dim3(blocks_for((int64_t)b * n * (d / 4))), dim3(256), 0, s, F4M(x), inv, F4(mask_token), F4(pos), b, n,
                       d / 4);
  else
    hipLaunchKernelGGL(simmim_select_kernel<1>, dim3(blocks_for((int64_t)b * n * d)), dim3(256), 0, s, x, inv, mask_token, pos, b, n, d);
}

void launch_zero_selected_rows(float* x, const int32_t* inv, int64_t rows, int d, hipStream_t s) {
  if (rows * d == 0) return;
  if (vec_ok(d, {x}))
    hipLaunchKernelGGL(zero_selected_rows_kernel<4>, dim3(blocks_for(rows * (d / 4))), dim3(256), 0, s, F4M(x), inv, rows, d / 4);
  else
    hipLaunchKernelGGL(zero_selected_rows_kernel<1>, dim3(blocks_for(rows * d)), dim3(256), 0, s, x, inv, rows, d);
}

void launch_scatter_by_index(const float* src, const int32_t* idx, int b, int k, int d, float* dst, int64_t dst_batch_stride, int row0, hipStream_t s) {
  const int64_t total = (int64_t)b * k * d;
  if (total == 0) return;
  hipLaunchKernelGGL(scatter_by_index_kernel, dim3(blocks_for(total)), dim3(256), 0, s, src, idx, b, k, d, dst, dst_batch_stride, row0);
}
void launch_mpp_labels(const float* img, int b, int H, int W, int C, int p, int bits, float mpv, int has_norm, const float* mean, const float* std_,
                       const int32_t* idx, int k, int32_t* labels, hipStream_t s) {
  if (b * k == 0) return;
  const float4 m4 = make_float4(mean[0], mean[1], mean[2], mean[3]), s4 = make_float4(std_[0], std_[1], std_[2], std_[3]);
  hipLaunchKernelGGL(mpp_labels_kernel, dim3(blocks_for((int64_t)b * k)), dim3(256), 0, s, img, H, W, C, p, bits, mpv, has_norm, m4, s4, idx, b, k, labels);
}

int64_t recon_loss_ws_elems(int64_t count) { return ceil_div(count, LOSS_THREADS * LOSS_PER_THREAD) + 1; }
void launch_recon_loss(const float* pred, const float* target_or_null, int64_t count, int kind, float scale, float* dpred, float* partial_ws,
                       float* loss_out, hipStream_t s) {
  const int nblk = (int)ceil_div(count, LOSS_THREADS * LOSS_PER_THREAD);
  hipLaunchKernelGGL(recon_loss_kernel, dim3((unsigned)nblk), dim3(LOSS_THREADS), 0, s, pred, target_or_null, count, kind, scale, dpred, partial_ws);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(64), 0, s, partial_ws, nblk, loss_out);
}
