// Host-side engine behind the C ABI: parameter table, device memory plan, and the launch sequences
// that replace ViT.call / DeepViT.call / CaiT.call (vit.py:159-177, deepvit.py:139-157,
// cait.py:180-194) and their autodiff.
#pragma once
#include <string>
#include <vector>

#include "../../include/vitx.h"
#include "kernels.h"

struct ParamDesc {
  std::string name;
  std::vector<int64_t> shape;
  int64_t offset = 0, count = 0;   // offset in the packed host blob (API order)
  int64_t aoff = 0;                // offset in the device arena (every tensor starts 16-B aligned)
};

// Builds the explicit parameter order (must match oracle/spec.py:param_spec; tests compare them).
// Returns "" on success or the reference's assertion text on an invalid config.
std::string build_param_table(const vitx_config& c, std::vector<ParamDesc>& out);

// native data-parallel exchange (comm.hip)
struct CommState {
  hipStream_t stream = nullptr;      // the collectives' stream (high priority)
  void* all_reduce = nullptr;        // ncclAllReduce (dlsym)
  int overlap = 0;                   // 1: buckets are launched from inside the backward pass as they complete
  int64_t bucket = 0;                // elements per bucket (multiple of 4)
  int wire_bf16 = 0;
  bf16_t* wire = nullptr;            // [n_arena] bf16 wire copies (allocated on first use)
  std::vector<int64_t> covered;      // per bucket: elements reported final
  std::vector<char> launched;
  std::vector<hipEvent_t> ready_ev, done_ev;
  int last_launched = -1, n_launched = 0, last_overlapped = 0;
  int next_bucket = -1;              // the next bucket of the fixed (descending) launch order; -1 = not started (the last one), -2 = all out
  // (round 6) which Dense launches run beside a collective is decided in STREAM ORDER from the collectives' measured durations, not from a host-side
  // event query (the host enqueues a backward pass tens of milliseconds ahead of the GPU: "has the newest bucket finished" is almost never true
  // at enqueue time, so every launch behind the first bucket used to take the one-tile-per-workgroup form).  t0_ev / t1_ev[parity][bucket] bracket a
  // bucket's work on the communication stream; the previous-but-one exchange's brackets are read when the bucket is launched again.
  std::vector<hipEvent_t> t0_ev[2], t1_ev[2];
  std::vector<char> timed[2];        // brackets of that parity were recorded
  std::vector<float> coll_ms;        // last measured duration per bucket; < 0: not measured yet
  int parity = 0;                    // flips with every finished exchange
  int shared_credit = 0;             // Dense launches, from now on in stream order, that are taken to run beside a collective
  int64_t busy_hits = 0;
  std::string failed;                // error of a launch made from inside the backward (surfaced by the next vitx_allreduce_grads)
};

struct Dense {
  int in = 0, out = 0;
  int in_k = 0, out_k = 0;          // padded to 64 (K extents of the bf16 GEMMs)
  int64_t w = -1, b = -1;           // arena offsets (Keras kernel [in,out], bias [out])
  bf16_t* wt = nullptr;             // [round_up(out,256)][in_k]   forward operand  (W^T)
  bf16_t* wn = nullptr;             // [round_up(in,256)][out_k]   dgrad operand    (W)
  // a layer whose fp32 parameters / gradients live outside the engine's arenas (the MAE / SimMIM wrappers' own Dense layers)
  const float *ext_w = nullptr, *ext_b = nullptr;
  float *ext_gw = nullptr, *ext_gb = nullptr;
};

struct BlockParams {
  int64_t a_scale = -1, ln1_g = -1, ln1_b = -1;
  Dense qkv, q, kv, out;
  Dense qkvcat;   // CaiT patch stage, bf16 mode: operand copies of [to_q | to_kv] side by side (w = -1: no parameter of its own)
  bool has_out = true;
  int64_t mix_pre = -1, mix_post = -1, re_w = -1, re_g = -1, re_b = -1;
  int64_t m_scale = -1, ln2_g = -1, ln2_b = -1;
  Dense fc1, fc2;
  int64_t p_begin = 0, p_end = 0;   // arena range of this block's parameters
};

struct BlockActs {
  float *x_in = nullptr, *x_mid = nullptr, *x_out = nullptr;
  void *y1 = nullptr, *qkv = nullptr, *q = nullptr, *kv = nullptr, *ctx = nullptr, *o = nullptr, *fa = nullptr;
  void *y2 = nullptr, *hpre = nullptr, *act = nullptr, *fm = nullptr;
  float *mean1 = nullptr, *rstd1 = nullptr, *mean2 = nullptr, *rstd2 = nullptr, *lse = nullptr;
  // parallel branches (parallel_vit.py:36-42): a layer of P attention + P feed-forward blocks is laid out as 2P half-blocks.  Each
  // adds its branch to the running residual (x_in -> x_mid, or x_mid -> x_out) but normalises the LAYER's input (ln*_src) --
  // null = the residual input itself, i.e. the ordinary block.
  const float *ln1_src = nullptr, *ln2_src = nullptr;
  bool skip_attn = false, skip_mlp = false;
  // materialised-attention path (DeepViT / CaiT, and ViT in parity mode): the score tensors of the last forward of THIS block, kept
  // for its backward instead of recomputing QK^T and the head-axis chain (sized once for max_batch; HBM is 288 GB)
  float* sc_keep[3] = {nullptr, nullptr, nullptr};
  int64_t sc_keep_elems = 0;        // 0: not tried yet, -1: over the budget (this block recomputes), else elements per buffer
  int64_t sc_geom = -1;             // (b, nq, nk) the kept tensors describe
  int sc_pi = 0;                    // index of the tensor that multiplies V
  bool sc_no_mixed = false;         // kept by the one-kernel DeepViT forward: sc_keep[1] (the mixed scores) was not written
  bool sc_a2_bf16 = false;   // sc_keep[2] holds bf16 [nq][round_up(nk, 8)] planes (one-kernel forward with the one-kernel backward as its reader)
  int par_first = 0, par_last = 0;   // backward order inside a group of parallel half-blocks: first / last one processed (1 + 1 = alone)
};

struct Stage {
  std::string prefix;
  int depth = 0;
  int nq_max = 0, nc_max = 0;       // query tokens per image, extra context tokens per image
  std::vector<BlockParams> bp;
  std::vector<BlockActs> ba;
};

struct ProfEvent {
  int cls;
  int cls2 = -1;   // optional second class the same launch is booked under ("shape ..." rows: one per GEMM shape x epilogue)
  hipEvent_t e0, e1;
  double flops, bytes;
};

// Weight-gradient side stream (engine.hip, "side stream" section): a buffer the dgrad chain rewrites every block while a weight-gradient
// GEMM queued on the side stream may still be reading it.  The chain writes into the next slot of a small ring instead; a slot is reused
// only after the side-stream reader recorded on it has finished (an event wait on the main stream, normally already satisfied).
constexpr int SIDE_RING_MAX = 8;
struct SideRing {
  void* slot[SIDE_RING_MAX] = {};
  hipEvent_t rd[SIDE_RING_MAX] = {};
  bool pend[SIDE_RING_MAX] = {};
  int n = 0, cur = 0;
};
struct PendingReady { int64_t off, cnt; hipEvent_t ev; };

struct vitx_engine {
  vitx_config cfg{};
  std::vector<ParamDesc> table;
  int64_t n_params = 0;              // packed element count (host blob)
  int64_t n_arena = 0;               // device arena element count (>= n_params, zero padding between tensors)
  bool bf16 = false;
  bool x3_attn = false;
  bool x3_fused_attn = true;         // BF16X3: ViT attention through the fused split-operand kernel (attn_x3.hip) instead of materialised scores
  bool x3 = false;                   // BF16X3: the fp32 mode's data path with the GEMMs on split bf16 operands (gemm_bf16x3.hip)
  int esz = 4;                       // bytes per "T" element

  // derived sizes (configured image)
  int ntok_cap = 0;                  // rows per image the buffers hold: ntok_max plus one slot for a distillation token (distill.py:26-28)
  int np_max = 0, ntok_max = 0, pd = 0, pd_k = 0, inner = 0, nc_k = 0;
  int64_t mp = 0;                    // max token rows, padded to 256
  int64_t mpp = 0;                   // max patch rows, padded to 256
  int64_t bp = 0;                    // max batch padded to 256

  hipStream_t own_stream = nullptr, stream = nullptr;
  // weight gradients (no consumer until the optimizer / the gradient exchange) run on `side`, forked from / joined into `stream` by events
  hipStream_t side = nullptr;
  int side2_min_rows = 8192;     // VITX_LN_REDUCE_SIDE_ROWS (read once per handle): VJPs of fewer rows keep their reductions on the main stream
  hipStream_t side2 = nullptr;   // the small reductions of the LayerNorm VJPs: they must not queue behind a block's weight-gradient GEMMs
  int side_mode = 1;                 // VITX_SIDE_STREAM=0: everything on one stream (A/B reference)
  bool side_live = false;            // inside a backward pass that uses the side stream
  bool side_dirty = false;           // work was queued on the side stream since the last join
  SideRing rg_dh, rg_glp, rg_dqkv, rg_dbr, rg_lnp, rg_cs;
  float* cs_part = nullptr;   // per-tile column sums of d hpre (fc1 bias gradient): same treatment
  float* ln_part = nullptr;   // LayerNorm-VJP partial sums of the blocks, their own ring: the reduction runs on the side stream
  std::vector<hipEvent_t> side_events; size_t side_ev_next = 0;
  void* conv_descs = nullptr; int conv_n = 0, conv_blocks = 0;   // device table of the batched bf16 operand refresh (built on first use)
  std::vector<PendingReady> side_ready;   // gradient-ready reports waiting for the side stream's share of their range
  float* params = nullptr;
  float* grads = nullptr;
  bool own_params = true, own_grads = true;
  bool params_dirty = true;

  // PatchMerger (vit_with_patch_merger.py:42-55): merge_after = index of the layer it follows (-1: never), merge_t = tokens out
  int merge_after = -1, merge_t = 0;
  int64_t pm_g = -1, pm_b = -1, pm_q = -1;
  float *pm_xn = nullptr, *pm_mean = nullptr, *pm_rstd = nullptr, *pm_attn = nullptr, *pm_dattn = nullptr, *pm_out = nullptr, *pm_dxn = nullptr,
        *pm_dxn2 = nullptr, *pm_dq = nullptr;

  // parameter handles
  int64_t pos = -1, cls = -1, head_g = -1, head_b = -1;
  Dense patch, head;
  std::vector<Stage> stages;

  // buffers
  std::vector<void*> allocs;
  std::vector<std::pair<void*, size_t>> t_buffers;   // re-zeroed when the batch geometry changes
  int64_t ws_bytes = 0;
  float* img_dev = nullptr;
  void* patches = nullptr;
  float* pooled = nullptr; void* yh = nullptr; float *mean_h = nullptr, *rstd_h = nullptr;
  float *logits = nullptr, *dlogits = nullptr; void* dl_lp = nullptr; void* dyh = nullptr; float* dpooled = nullptr;
  float* g2 = nullptr;               // parallel branches: the layer-input gradient being accumulated while g still feeds the other branches
  float* g = nullptr; void* g_lp = nullptr; float* g_ctx = nullptr;  // residual gradient stream (+T copy), CaiT patch-output grad
  void *d_h = nullptr, *d_y = nullptr, *d_o = nullptr, *d_qkv = nullptr, *d_ctx = nullptr, *d_br = nullptr;
  bf16_t *xt = nullptr, *dyt = nullptr; int64_t t_rows = 0;
  float* partial_ws = nullptr; int64_t partial_elems = 0;
  float* red_ws = nullptr; int64_t red_elems = 0;
  float* sc[4] = {nullptr, nullptr, nullptr, nullptr}; int64_t sc_elems = 0;
  float* dsum = nullptr;
  void* zero_page = nullptr;         // 256 B of zeros (source of the padded rows in the attention DMA staging)
  float* tmp_f32 = nullptr;          // [mp, max(d, pd)] fp32 scratch (dropout / dimg paths)
  float* loss_rows = nullptr;
  float* distill_ws = nullptr;       // [1 + max_batch, dim]: distillation token / its gradient (row 0) and the per-image tokens / their cotangents (host entry points)
  float *opt_m = nullptr, *opt_v = nullptr; int opt_step = 0;   // optimizer state (allocated on first use)
  bf16_t *bench_a = nullptr, *bench_b = nullptr; float* bench_c = nullptr; int64_t bench_elems = 0;

  // state of the last forward
  bool have_fwd = false;
  bool have_tf = false;              // saved activations describe a transformer_forward(tokens) of [tf_b, tf_n, dim]
  int next_patch_np = 0;                // vitx_set_patch_input: the next host-pointer forward entry reads patch rows [b, np, pd] (one shot)
  const float* fwd_patches = nullptr;   // set by the forward_patches entry points for the next engine_forward (consumed there)
  int fwd_np = 0;
  bool last_from_patches = false;
  int tf_b = 0, tf_n = 0;
  float tf_drop = 0.f;               // dropout rate that transformer_forward applied (0 unless training) and the seed of its masks
  uint64_t tf_seed = 0;
  bool have_embed = false, have_head = false;   // efficient.ViT shell: state of the last embed_forward / head_forward
  int shell_b = 0, shell_n = 0;
  float* shell_x = nullptr;          // [mp, dim] fp32 copy of the head's input (allocated on first use)
  bool have_pt = false;              // e->patches holds the unfolded patches of a patch_tokens_forward of [pt_b, pt_np] patches
  int pt_b = 0, pt_np = 0;
  int64_t patch_rows = -1;           // rows of e->patches written by the last unfold (rows beyond it are zero)
  void* pt_dy = nullptr;             // [mpp, dim] T copy of d(tokens) for the patch-embedding weight gradient (allocated on first use)
  int64_t pt_dy_rows = -1;
  int last_extra = 0;                // 1: the last forward carried a distillation token as its last row (last_ntok includes it)
  int last_b = 0, last_np = 0, last_ntok = 0, last_H = 0, last_W = 0, last_training = 0;
  uint64_t last_seed = 0;
  std::vector<std::vector<bool>> layer_kept;   // per stage: blocks that survived CaiT layer dropout in the last forward
  int64_t zero_geom = -1;

  bool keep_scores = true;           // VITX_RECOMPUTE_SCORES=1: recompute in the backward as before (A/B)
  int64_t sc_keep_bytes = 0, sc_keep_budget = 48LL << 30;
  // env switches
  bool force_generic_gemm = false, force_generic_attn = false, wgrad_via_transpose = true;
  int gemm_kernel = 0;
  int gemm_tail = 0;      // forced tail variant (VITX_GEMM_TAIL_KERNEL; only with a forced main variant)
  int reverse_mask = 7;                  // VITX_REVERSE=bits: 1 forward GEMMs, 2 dgrad GEMMs walk their row tiles last-to-first when the A operand exceeds
  int64_t reverse_min_bytes = 200ll << 20;   //   VITX_REVERSE_MIN_MB (default 200 MB), see decode_tile (gemm_bf16.hip); 4: the fused attention forward (attn_bf16.hip)
  bool bgemm_pairs = true;               // VITX_BGEMM_PAIRS=0: the four batched products of the materialised attention backward as four launches
  int nt_mask = 17;                      // VITX_NT=bits: non-temporal hints.  1: the fc1 epilogue's gelu'(h) store (read again only by the backward);
                                         // 16: the fc1- and qkv-dgrad outputs d(y), read by the LayerNorm backward only after the weight gradient that
                                         // re-reads d(hpre) / d(qkv) (TN class -0.25 ms per step).
                                         // (Measured and dropped: the same hint on the fc2-dgrad epilogue's read of it -- no effect -- and on the
                                         // weight-gradient operand loads -- 10.8 -> 11.25 ms per step.)
  bool mlp_bwd_consumers_first = true;   // VITX_MLP_BWD_ORDER=0: fc2 weight gradient between the producer and the consumers of d hpre
  bool deepvit_fused_bwd = true;     // VITX_DEEPVIT_FUSED_BWD=0: the backward of that kernel as batched GEMMs + point / row kernels (A/B reference)
  bool glp_skip = true;             // LayerNorm VJPs skip the bf16 copy of the residual gradient when no branch reads it (all blocks have LayerScale); VITX_GLP_SKIP=0: always written
  bool ln_scale_fused = true;       // CaiT: a LayerNorm VJP also runs the LayerScale VJP of the branch that consumes its result (VITX_LN_SCALE_FUSED=0: a pass of its own)
  int64_t dbr_ready = 0;            // branch_key of the branch whose gradient e->d_br already holds (set by that LayerNorm VJP, cleared by the branch)
  bool score_bf16 = true;           // score tensors whose only reader is a batched product are kept / written as bf16 by the one-kernel attention paths (VITX_SCORE_BF16=0: fp32)
  bool cait_qkv_cat = true;         // CaiT patch stage: to_q and to_kv as one Dense on concatenated operand copies (VITX_CAIT_QKV_CAT=0: two launches each way)
  bool cait_fused = true;           // cait.py:121-128 forward as one kernel in the bf16 mode (attn_cait_fused.hip); VITX_CAIT_FUSED=0 disables
  bool deepvit_fused = true;         // VITX_DEEPVIT_FUSED=0: DeepViT attention forward as batched GEMMs + head-axis kernels (A/B reference)
  bool unfused_headops = false;      // VITX_UNFUSED_HEADOPS=1: separate mix / softmax / LayerNorm-over-heads kernels (A/B reference)
  int gemm_stagger = 0;

  // profiling
  bool profiling = false;
  std::vector<ProfEvent> prof_events;
  std::vector<std::string> prof_names;

  // data parallel
  vitx_grad_ready_fn grad_cb = nullptr; void* grad_cb_user = nullptr;
  void* rccl_lib = nullptr; void* comm = nullptr; int rank = 0, world = 1;
  CommState cm;
};

// comm.hip: the library's own gradient exchange (RCCL, dlopen'ed)
int comm_unique_id(void* out128, std::string& err);
int comm_init(vitx_engine* e, int rank, int world, const void* uid, std::string& err);
int comm_overlap(vitx_engine* e, int enable, int64_t bucket_bytes, int wire_bf16, std::string& err);
void comm_on_ready(vitx_engine* e, int64_t off, int64_t cnt);
int comm_busy(vitx_engine* e);
void comm_wait_event(vitx_engine* e, hipEvent_t ev);   // the communication stream waits for `ev` (side-stream share of a reported range)
int comm_finish(vitx_engine* e, std::string& err);
void comm_stats(vitx_engine* e, int64_t* out4);
void comm_destroy(vitx_engine* e);

int engine_create(const vitx_config& cfg, vitx_engine** out, std::string& err);
void engine_destroy(vitx_engine* e);
// distill_token_dev [dim] (optional): DistillMixin.call (distill.py:16-44) -- the token is appended after the position embedding,
// attended with the rest, split off before pooling and returned per image in distill_out_dev [b, dim]
int engine_forward(vitx_engine* e, const float* img_dev, int b, int H, int W, int training, uint64_t seed, float* logits_dev,
                   std::string& err, const float* distill_token_dev = nullptr, float* distill_out_dev = nullptr);
// d_distill_dev [b, dim]: cotangent of distill_out; d_token_out_dev [dim]: gradient of the distillation token
int engine_backward(vitx_engine* e, const float* dlogits_dev, float* dimg_dev, std::string& err, const float* d_distill_dev = nullptr,
                    float* d_token_out_dev = nullptr);
int engine_transformer_forward(vitx_engine* e, const float* tokens_dev, int b, int n, int training, uint64_t seed, float* out_dev, std::string& err);
int engine_transformer_backward(vitx_engine* e, const float* dout_dev, float* dtokens_dev, std::string& err);
int engine_patch_tokens_forward(vitx_engine* e, const float* img_dev, int b, int H, int W, float* tokens_dev, float* patches_f32_dev_or_null,
                                std::string& err);
int engine_patch_tokens_backward(vitx_engine* e, const float* dtokens_dev, std::string& err);
// efficient.ViT shell (efficient.py:12-56): embedding in front of / pooling + mlp_head behind a caller-supplied transformer
int engine_embed_forward(vitx_engine* e, const float* img_dev, int b, int H, int W, float* tokens_dev, std::string& err);
int engine_patch_dense_forward(vitx_engine* e, const float* patches_dev, int rows, float* out_dev, std::string& err);
int engine_head_forward(vitx_engine* e, const float* x_dev, int b, int n, float* logits_dev, std::string& err);
int engine_head_backward(vitx_engine* e, const float* dlogits_dev, float* dx_dev, std::string& err);
int engine_embed_backward(vitx_engine* e, const float* dtokens_dev, float* dimg_dev, std::string& err);
void engine_refresh_weights(vitx_engine* e);
void engine_params_moved(vitx_engine* e);   // the parameter arena pointer changed (vitx_bind_arenas): re-point the batched operand refresh
// Dense layers owned by a wrapper object but run with this engine's GEMM kernels, workspaces and stream (bf16 mode: X / dY are
// row-padded bf16 buffers as everywhere else in the engine; parity mode: fp32).  y / dx are fp32 [rows, out] / [rows, in].
int engine_ext_dense_init(vitx_engine* e, Dense& w, int in, int out, const float* W, const float* bias, float* gW, float* gbias, std::string& err);
void engine_ext_dense_refresh(vitx_engine* e, const Dense& w);
void engine_ext_dense_fwd(vitx_engine* e, const void* X, int64_t ldx, int rows, const Dense& w, float* y);
void engine_ext_dense_bwd(vitx_engine* e, const void* X, int64_t ldx, const void* dY, int64_t ldy, const float* dY_f32, int rows, const Dense& w,
                          float* dx_or_null);
int engine_check_gemm(vitx_engine* e, int kind, int M, int N, int K, int kernel, int epilogue, float* errs, std::string& err);
int engine_bench_gemm(vitx_engine* e, int M, int N, int K, int kernel, int epilogue, int iters, float* avg_ms, float* max_err,
                      std::string& err);
int prof_class(vitx_engine* e, const char* name);
