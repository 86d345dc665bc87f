// Host-callable launchers of every HIP kernel in libvitx (one translation unit per kernel family).
#pragma once
#include "common.h"
#include "epilogue.h"

// ---------------------------------------------------------------- gemm_generic.hip
struct GenericGemmArgs {
  const void* A = nullptr;
  const void* B = nullptr;
  int M = 0, N = 0, K = 0;
  int64_t sam = 0, sak = 1, sbk = 0, sbn = 1;
  int nb = 1, nh = 1;
  int64_t sAb = 0, sAh = 0, sBb = 0, sBh = 0;
  int k_last = 0;   // > 0: K extent of the LAST batch slice (split-K over the batch index: slices of K rows, the last one shorter); gemm_bf16x3.hip only
  int x3 = 0;   // BF16X3 compute mode: all-fp32 problems run as three bf16 MFMA products of hi / lo split operands (gemm_bf16x3.hip)
};
void launch_gemm_generic(const GenericGemmArgs& g, const EpiParams& ep, int mode, int ta, int tb, int to, hipStream_t s);
// ---------------------------------------------------------------- gemm_f32_mfma.hip
// the same contract on v_mfma_f32_32x32x2_f32 for all-fp32 problems (bit-identical to the scalar kernel: a k-ordered fmaf chain);
// launch_gemm_generic routes to it when supported (VITX_F32_MFMA=0 keeps the scalar kernel)
bool gemm_f32_mfma_supported(const GenericGemmArgs& g, int ta, int tb, int to);
void launch_gemm_f32_mfma(const GenericGemmArgs& g, const EpiParams& ep, int mode, hipStream_t s);
void gemm_f32_mfma_read_env();   // VITX_F32_MFMA, read once per engine handle (not per launch)
// ---------------------------------------------------------------- gemm_bf16x3.hip
// the same contract with every fp32 operand split into bf16 hi + lo and three bf16 MFMA products per k-step (~2^-16 per product);
// launch_gemm_generic routes to it when GenericGemmArgs::x3 is set
bool gemm_bf16x3_supported(const GenericGemmArgs& g, int ta, int tb, int to);
void launch_gemm_bf16x3(const GenericGemmArgs& g, const EpiParams& ep, int mode, hipStream_t s);

// ---------------------------------------------------------------- attn_headchain.hip
// fused head-axis chains (one wave per (image, query) row): CaiT talking heads, DeepViT re-attention, and their VJPs
bool headchain_supported(int h, int nk);
int64_t headchain_ws_elems(int h);
void launch_cait_chain_fwd(const float* s0, const float* wpre, const float* wpost, float* a1_or_null, float* a2, int b, int h, int nq, int nk,
                           int64_t ld, hipStream_t s);
void launch_cait_chain_bwd(const float* s0, const float* a1, float* da_inout, const float* wpre, const float* wpost, float* ws, float* dwpre,
                           float* dwpost, int b, int h, int nq, int nk, int64_t ld, hipStream_t s, bf16_t* ds_lp = nullptr);
bool cait_chain_bwd_bf16_out_ok();
void launch_deepvit_chain_fwd(float* s0_inout, const float* wre, const float* gamma, const float* beta, float* mixed_or_null, float* a2, int keep,
                              int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s);
void launch_deepvit_chain_bwd(const float* a0, const float* mixed, float* da_inout, const float* wre, const float* gamma, float* ws, float* dwre,
                              float* dgamma, float* dbeta, int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s);

// ---------------------------------------------------------------- attn_bgemm_mfma.hip
// batched small GEMM (M, N, K <= 128 per (image, head)) on MFMA for the materialised attention path in bf16 mode
bool bgemm_mfma_supported(const GenericGemmArgs& g, int ta, int tb, int to, int mode);
void launch_bgemm_mfma(const GenericGemmArgs& g, const EpiParams& ep, int ta, int to, hipStream_t s);
void launch_bgemm_mfma_pair(const GenericGemmArgs& g1, const EpiParams& ep1, int ta1, int to1, const GenericGemmArgs& g2, const EpiParams& ep2, int ta2,
                            int to2, hipStream_t s);   // two products of the same (image, head) back to back in one launch (same nb, nh)

// ---------------------------------------------------------------- gemm_bf16.hip
// C[M,N] = A[M,K] * B[N,K]^T, bf16 operands (K contiguous), fp32 MFMA accumulation.
struct Bf16GemmArgs {
  const bf16_t* A = nullptr;
  const bf16_t* B = nullptr;
  int64_t lda = 0, ldb = 0;
  int M = 0, N = 0, K = 0;   // K multiple of 64; A has >= round_up(M,tile) rows, B >= round_up(N,tile) rows
  int split_k = 1;           // >1: EPI_PARTIAL slices of K/split_k (each a multiple of 64)
  int kernel = 0;            // tile variant: 0 auto, 1 = 128x128 (4 waves), 2 = 256x256 (8 waves), 3 = 256x128
  unsigned long long* stamps = nullptr;   // diagnostics (VITX_GEMM_STAMPS=1 in vitx_bench_gemm): cycle stamps of the tile phases, [256 WGs][16 tiles][4]
  int phase = 0;             // persistent pipelined kernels: workgroup i starts ((i >> 3) & 7) * phase * 4096 cycles late (de-phases the tile loops)
  int reverse_m = 0;         // 1: row tiles are walked from the last to the first (see decode_tile)
  int shared_gpu = 0;        // 1: a collective of this handle may hold CUs while this launch runs (comm_busy): one tile per workgroup instead of a persistent grid
  int walk = 0;              // set by the launcher of the pipelined kernel: 2 = XCD-owned row bands (see gemm_bf16_nt_pipe_kernel)
  int stagger = 0;           // >0: first-wave workgroups start (cu_slot & 3) * stagger * 2048 cycles late (de-phases store-heavy epilogues)
  int tail = 0;              // (round 6) tail balancing: > 0 = tile variant (1 = 128 x 128, 3 = 256 x 128, 10 = 192 x 128) for the rows of the last PARTIAL round of the
                             //   persistent grid, launched on their own behind the full rounds (gemm_bf16.hip: dispatch_gemm_bf16_tail); 0 = one launch
};
void launch_gemm_bf16(const Bf16GemmArgs& g, const EpiParams& ep, int mode, hipStream_t s);
int gemm_bf16_tile_m(int kernel, int M, int N);
int gemm_bf16_tile_n(int kernel, int M, int N);
int gemm_bf16_num_slices(int K, int split_k);
void gemm_bf16_set_shared_gpu(int on);   // 1: other kernels (collectives) share the CUs -> no persistent variants
void gemm_bf16_allow_320(int on);   // the 320-row tile needs operand/output buffers with 320 spare rows
// weight-gradient form: C[M,N] partials = A[K,M]^T * B[K,N] (A, B row-major with leading dims lda, ldb; K = token rows, multiple of 64)
void launch_gemm_bf16_tn(const Bf16GemmArgs& g, const EpiParams& ep, hipStream_t s);
int gemm_bf16_tn_tile(int kernel, int M, int N);

// ---------------------------------------------------------------- attn_bf16.hip
// fused multi-head self-attention (vit.py:73-82) on packed qkv [b, n, 3, h, 64] bf16.
bool attn_bf16_supported(int n, int dim_head);
void launch_attn_bf16_fwd(const bf16_t* qkv, bf16_t* o, float* lse, int b, int n, int h, float scale, const bf16_t* zero_page, int reverse,
                          hipStream_t s);   // zero_page: >= 128 B of zeros; reverse: 1 = (image, head) tasks from the last to the first
// attn_x3.hip: the same fused attention on fp32 storage with split-operand (hi + lo) bf16 MFMA products (BF16X3 mode)
bool attn_x3_supported(int n, int dim_head);
void launch_attn_x3_fwd(const float* qkv, float* o, float* lse, int b, int n, int h, float scale, hipStream_t s);
void launch_attn_x3_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, int b, int n, int h, float scale, hipStream_t s);
void launch_attn_bf16_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* d_o, const float* lse, float* dsum_ws,
                          bf16_t* dqkv, int b, int n, int h, float scale, const bf16_t* zero_page, hipStream_t s);

// ---------------------------------------------------------------- elementwise.hip
void launch_unfold(const float* img, void* out, int out_bf16, int b, int H, int W, int C, int ph, int pw,
                   int64_t ldo, hipStream_t s);
void launch_fold_add(const float* dpatches, int64_t ld, float* dimg, int b, int H, int W, int C, int ph, int pw, hipStream_t s);
// tf.image.extract_patches(..., rates 1, padding 'SAME') (t2t.py:42) and its VJP; out [b, ceil(H/st), ceil(W/st), k*k*C]
void extract_patches_geometry(int H, int W, int k, int st, int* oh, int* ow, int* pad_top, int* pad_left);
void launch_extract_patches(const float* x, float* out, int b, int H, int W, int C, int k, int st, hipStream_t s);
void launch_extract_patches_bwd(const float* dout, float* dx, int b, int H, int W, int C, int k, int st, hipStream_t s);
void launch_cls_pos_row(float* x, const float* cls, const float* pos, int b, int ntok, int d, int64_t ldx, hipStream_t s);
void launch_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int y_bf16,
                          int64_t ldy, float* mean, float* rstd, int rows, int d, float eps, hipStream_t s);
// g_out = (g_in ? g_in : 0) + LN_bwd(dy); optional low-precision copy of g_out; dgamma/dbeta via partial_ws
void launch_layernorm_bwd(const void* dy, int dy_bf16, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                          const float* rstd, const float* gamma, const float* g_in, int64_t ldgi, float* g_out, int64_t ldgo,
                          void* g_lp, int64_t ldglp, float* partial_ws, float* dgamma, float* dbeta, float* gsum, int rows, int d,
                          hipStream_t s, int* deferred = nullptr);   // gsum (optional): column sums of g_in; deferred: see the definition
void launch_layernorm_bwd_reduce(float* partial_ws, int nparts, int d, float* dgamma, float* dbeta, float* gsum, hipStream_t s);
int64_t layernorm_bwd_ws_elems(int d);
// LayerNorm VJP that also runs the LayerScale VJP of the branch consuming its result (bf16 mode, CaiT): dbr = g_out * nscale, partial rows for
// d nscale = colsum(g_out * nf) and that branch's bias gradient nscale * colsum(g_out); the caller runs the reduction on a stream of its choice
bool layernorm_bwd_scale_ok(int d);   // (every operand 16-B aligned with row strides that are multiples of 4: true of the engine's own buffers)
void launch_layernorm_bwd_scale(const bf16_t* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                                const float* g_in, int64_t ldgi, float* g_out, int64_t ldgo, bf16_t* dbr, int64_t lddbr, const bf16_t* nf, int64_t ldnf,
                                const float* nscale, float* partial_ws, int rows, int d, hipStream_t s, int* nparts);
void launch_layernorm_bwd_scale_reduce(float* partial_ws, int nparts, int d, float* dgamma, float* dbeta, float* dscale, float* dbias, const float* nscale,
                                       hipStream_t s);
void launch_colsum(const void* x, int is_bf16, int64_t ld, int rows, int cols, float* partial_ws, float* out, hipStream_t s);
int64_t colsum_ws_elems(int cols);
void launch_reduce_partials(const float* partial, int nparts, int64_t stride, int64_t n, float* out, float alpha, hipStream_t s);
void launch_reduce_partials3(const float* partial, int nparts, int64_t stride, int64_t seg, int nseg, float* out0, float* out1, float* out2,
                             float* ws2, float alpha, hipStream_t s);
void launch_convert_weight(const float* w, int in, int out, bf16_t* wn, int64_t ldwn, bf16_t* wt, int64_t ldwt, hipStream_t s);
// one table entry per Dense kernel; block0 = first block of this matrix in the batched launch (tiles of 64 x 64, tiles_x = ceil(out / 64))
struct ConvertDesc { const float* w; bf16_t* wn; bf16_t* wt; int64_t ldwn, ldwt; int in, out, tiles_x, block0; };
void launch_convert_weights_batched(const ConvertDesc* descs_dev, int n, int total_blocks, hipStream_t s);
void launch_transpose_bf16(const bf16_t* in, int64_t ldi, int rows, int cols, bf16_t* out, int64_t ldo, hipStream_t s);
void launch_convert(const float* in, int64_t ldi, void* out, int out_bf16, int64_t ldo, int rows, int cols, int64_t out_cols_zero_to,
                    hipStream_t s);
void launch_to_f32(const void* in, int in_bf16, int64_t ldi, float* out, int64_t ldo, int rows, int cols, hipStream_t s);
void launch_set_token_row(float* x, const float* tok, int b, int ntok, int row, int d, hipStream_t s);   // x[b, row, :] = tok[:]
void launch_mean_pool(const float* x, int b, int ntok, int d, float* out, hipStream_t s, int stride_tok = 0);   // stride_tok: rows per image in memory (0 = ntok)
void launch_mean_pool_bwd(const float* dp, int b, int ntok, int d, float* g, hipStream_t s, int stride_tok = 0);   // rows >= ntok of an image are left alone
void launch_batch_reduce(const float* g, int b, int ntok, int d, int j0, int nj, float* out, hipStream_t s);  // out[j][c] = sum_b g[b][j0+j][c]
void launch_extract_rows(const float* g, int b, int ntok, int tok_off, int np, int d, void* out, int out_bf16, int64_t ldo, hipStream_t s);
void launch_sum_rows(const float* in, int rows, int d, float* out, hipStream_t s);  // out[c] = sum_r in[r][c] (small)
void launch_ce_grad(const float* logits, int64_t ld, const int32_t* labels, int b, int nc, float inv_batch, float* dlogits,
                    float* loss, hipStream_t s);
void launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, int step, hipStream_t s);
void launch_sgd(float* p, const float* g, float* m, int64_t n, float lr, float mom, float wd, hipStream_t s);   // m may be null (plain SGD)
void launch_fill_zero(void* p, int64_t bytes, hipStream_t s);   // bytes multiple of 16
void launch_fill_random_bf16(bf16_t* p, int64_t n, uint32_t seed, float scale, hipStream_t s);
// *out_max (zero it first) = max(*out_max, max |a - b| / (1 + |b|)) over [rows, cols]; +inf when a NaN is met (GEMM self-checks at full size)
void launch_max_rel_diff(const void* a, const void* b, int is_bf16, int64_t rows, int cols, int64_t lda, int64_t ldb, float* out_max, hipStream_t s);
void launch_dropout(void* x, int is_bf16, int64_t n, float rate, uint64_t seed, uint32_t site, hipStream_t s);
void launch_axpy_resid(const float* resid, const float* f, const float* scale, float* out, void* keep, int keep_bf16, int64_t rows, int d,
                       hipStream_t s);   // out = resid + f*scale[col]; keep[T] = f
void launch_branch_grad(const float* g, const float* scale, void* out, int out_bf16, int64_t rows, int d, float rate, uint64_t seed,
                        uint32_t site, hipStream_t s);   // out[T] = dropout_mask(g*scale[col])
void launch_resid_add(const float* resid, const void* t, int t_bf16, float* out, int64_t n, hipStream_t s);  // out = resid + t

// ---------------------------------------------------------------- attn_generic.hip (materialised attention pieces)
void launch_softmax_rows(float* sc, int64_t rows, int n, int64_t ld, hipStream_t s);
void launch_softmax_bwd_rows(const float* p, float* dp_inout, int64_t rows, int n, int64_t ld, hipStream_t s);
// DeepViT forward chain (softmax -> re-attention mix -> LayerNorm over heads) as row statistics + one fused point kernel
int64_t deepvit_point_bwd_ws_elems(int h);
void launch_deepvit_point_bwd(const float* a0, float* da_inout, const float* w, const float* gamma, float* ws, float* dw,
                              float* dgamma, float* dbeta, int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s);
// attn_deepvit_fused.hip: the whole Re-attention forward (deepvit.py:79-88) in one kernel, bf16 mode, nk <= 80
bool deepvit_attn_fused_supported(int h, int dim_head, int nq, int nk);
// cait.py:121-128 forward in one kernel (attn_cait_fused.hip): raw scaled scores / softmax / mixed softmax are written (fp32 [b][h][nq][ld]) only when keep
bool cait_attn_fused_supported(int h, int dim_head, int nq, int nk);
void launch_cait_attn_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv, int64_t qb, int64_t kb, int64_t vb,
                          bf16_t* o, int64_t ldo, int64_t ob, const float* wpre, const float* wpost, float* s0_keep, float* a1_keep, float* a2_keep,
                          int keep, int b, int h, int nq, int nk, int64_t ld, float scale, const bf16_t* zero_page, hipStream_t s);
// the VJP of the same chain up to d(q) in one kernel (d(dots) leaves as fp32 for the d(k) product; P = what the fused forward kept)
int64_t deepvit_attn_bwd_ws_elems(int b, int h, int nq);
void launch_deepvit_attn_bwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv, int64_t qb, int64_t kb, int64_t vb,
                             const bf16_t* d_o, int64_t ldo, int64_t ob, const float* p_keep, const float* w, const float* gamma, float* ds_out,
                             bf16_t* dq, int64_t lddq, int64_t dqb, float* ws, float* dw, float* dgamma, float* dbeta, int b, int h, int nq, int nk,
                             int64_t ld, float scale, float eps, const bf16_t* zero_page, hipStream_t s, int ds_bf16 = 0);
void launch_deepvit_attn_fwd(const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv, int64_t qb,
                             int64_t kb, int64_t vb, bf16_t* o, int64_t ldo, int64_t ob, const float* w, const float* gamma,
                             const float* beta, float* p_keep, float* a2_keep, int keep, int b, int h, int nq,
                             int nk, int64_t ld, float scale, float eps, const bf16_t* zero_page, hipStream_t s);
bool deepvit_point_fwd_supported(int h, int nk);
int64_t deepvit_point_ws_elems(int b, int h, int nq);
void launch_deepvit_point_fwd(float* s0_inout, float* stats_ws, const float* w, const float* gamma, const float* beta, float* mixed, float* a2,
                              int keep, int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s);
// talking-heads style mix over the head axis of [b, h, nq, ld]: out[b,g,i,j] = sum_h in[b,h,i,j] * W[h,g]
void launch_headmix_fwd(const float* in, const float* w, float* out, int b, int h, int nq, int nk, int64_t ld, hipStream_t s);
// din[b,h,i,j] = sum_g dout[b,g,i,j] W[h,g];  dW[h,g] = sum_{b,i,j} in[b,h,i,j] dout[b,g,i,j]
void launch_headmix_bwd(const float* in, const float* dout, const float* w, float* din, float* dw_partial_ws, float* dw,
                        int b, int h, int nq, int nk, int64_t ld, hipStream_t s);
int64_t headmix_ws_elems(int b, int h, int nq, int nk);
// LayerNorm over the head axis at every (b,i,j)  (deepvit.py:59-63)
void launch_headnorm_fwd(const float* in, const float* gamma, const float* beta, float* out, int b, int h, int nq, int nk,
                         int64_t ld, float eps, hipStream_t s);
void launch_headnorm_bwd(const float* in, const float* dout, const float* gamma, float* din, float* partial_ws, float* dgamma,
                         float* dbeta, int b, int h, int nq, int nk, int64_t ld, float eps, hipStream_t s);
// ctx[b, 0:nq] = y[b]; ctx[b, nq:nq+nc] = context[b]   (cait.py:109-112), T-typed output
void launch_concat_ctx(const void* y, int y_bf16, const float* context, void* ctx, int ctx_bf16, int b, int nq, int nc, int d,
                       hipStream_t s);
// split d(ctx) [b, nq+nc, d] (T) back: dy_add[b, 0:nq] (T, overwritten) and dcontext[b, 0:nc] += (fp32 accumulate)
void launch_split_ctx_bwd(const void* dctx, int is_bf16, void* dy, float* dcontext, int b, int nq, int nc, int d, hipStream_t s);
void launch_add_T(void* a_inout, const void* b_in, int is_bf16, int64_t n, hipStream_t s);
void launch_scale_grad(const void* fx, int is_bf16, int64_t ldf, const float* g, int64_t ldg, int rows, int d, float* partial_ws,
                       float* dscale, hipStream_t s, const float* scale_or_null = nullptr, void* out_or_null = nullptr, int64_t ldo = 0,
                       float* dbias_or_null = nullptr);   // dbias (with out): += nothing, = column sums of out (the bias gradient of the Dense in front of the LayerScale)  // dscale[c] = sum_r g[r][c]*fx[r][c]   (cait.py:47-48 VJP)
void launch_mul_scale(const float* g, int64_t ldg, const float* scale, void* out, int out_bf16, int64_t ldo, int rows, int d,
                      hipStream_t s);          // out[T] = g * scale[col]
void launch_broadcast_rows(const float* src, int d, float* dst, int rows, hipStream_t s);  // dst[r][:] = src[:]

// ---------------------------------------------------------------- mim_ops.hip (MAE / SimMIM index, masking and loss kernels)
// idx: int32 [b, ldi]; inv: int32 [b, n] with inv[b, idx[b, j]] = j for j in [j0, j1), -1 elsewhere
void launch_index_inverse(const int32_t* idx, int64_t ldi, int b, int j0, int j1, int n, int32_t* inv, hipStream_t s);
// out[b, j, :] = src[b * src_batch_stride + idx[b, j0 + j] * d + :], j < k   (src_batch_stride = 0: one table for every image)
void launch_gather_rows(const float* src, int64_t src_batch_stride, const int32_t* idx, int64_t ldi, int j0, int b, int k, int d, float* out,
                        hipStream_t s);
// dst[b, t, :] = inv[b, t] in [j0, j1) ? src[b, inv[b, t] - j0, :] : 0        (src: [b, k_src, d], dst: [b, n, d])
void launch_scatter_rows(const float* src, int k_src, const int32_t* inv, int j0, int j1, int b, int n, int d, float* dst, hipStream_t s);
// dtab[t, :] (= | +=) sum_b (inv[b, t] in [j0, j1) ? src[b, inv[b, t] - j0, :] : 0)
void launch_table_grad(const float* src, int k_src, const int32_t* inv, int j0, int j1, int b, int n, int d, int accumulate, float* dtab,
                       hipStream_t s);
void launch_mae_assemble(const float* proj, const float* mask_token, const float* dpos, const int32_t* idx, int b, int np, int nm, int d, float* out,
                         hipStream_t s);
void launch_select_rowsum(const float* x, const int32_t* inv_or_null, int j0, int j1, int b, int n, int d, float* partial_ws, float* out,
                          hipStream_t s);   // partial_ws: b * d floats
void launch_simmim_select(float* x, const int32_t* inv, const float* mask_token, const float* pos, int b, int n, int d, hipStream_t s);
void launch_zero_selected_rows(float* x, const int32_t* inv, int64_t rows, int d, hipStream_t s);
void launch_scatter_by_index(const float* src, const int32_t* idx, int b, int k, int d, float* dst, int64_t dst_batch_stride, int row0, hipStream_t s);
void launch_mpp_labels(const float* img, int b, int H, int W, int C, int p, int bits, float mpv, int has_norm, const float* mean, const float* std_,
                       const int32_t* idx, int k, int32_t* labels, hipStream_t s);
int64_t recon_loss_ws_elems(int64_t count);
// kind 0: squared, 1: absolute; loss_out = scale * sum f(pred - target), dpred = scale * f'(pred - target); target may be null (= 0)
void launch_recon_loss(const float* pred, const float* target_or_null, int64_t count, int kind, float scale, float* dpred, float* partial_ws,
                       float* loss_out, hipStream_t s);
