/*
 * vitx.h -- C ABI of libvitx.so, the MI355X (gfx950) ViT forward/backward engine.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json: everything the
 * reference's Python classes hand to TensorFlow is handed to this library instead.
 * The reference exposes no FFI of its own (it is 24 files of Keras model definitions);
 * each entry point below cites the reference interface it replaces (paths relative to
 * /root/reference).  Plain C: pointers and sizes only, no C++/torch types, no exceptions.
 *
 * Conventions
 *   - every function returns int32 status: 0 = VITX_OK, <0 = error; the message of the last
 *     failure on the calling thread is available from vitx_last_error().
 *   - host buffers are borrowed for the duration of the call; the library owns all device memory
 *     (params, grads, saved activations, workspaces) unless arenas are bound with vitx_bind_arenas.
 *   - "_dev" variants take device pointers (inputs already resident in HBM) and are asynchronous
 *     on the handle's stream; the host-pointer variants synchronise before returning.
 *   - one handle <-> one GPU <-> one host thread at a time; different handles may be driven from
 *     different threads / processes (data parallel = one process per GPU, one handle each).
 *   - tensors are row-major fp32 at the boundary; images are NHWC like the reference's
 *     tf.random.normal([b, H, W, 3]) (vit_tensorflow/vit.py:193).
 */
#ifndef VITX_H_
#define VITX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITX_OK 0
#define VITX_ERR_INVALID (-1)      /* bad argument / config (mirrors the reference's AssertionError sites) */
#define VITX_ERR_HIP (-2)          /* HIP runtime failure */
#define VITX_ERR_UNSUPPORTED (-3)  /* valid in the reference, not yet handled by the engine */
#define VITX_ERR_STATE (-4)        /* call order (e.g. backward without forward) */
#define VITX_ERR_COMM (-5)         /* RCCL failure */

enum { VITX_VARIANT_VIT = 0, VITX_VARIANT_DEEPVIT = 1, VITX_VARIANT_CAIT = 2, VITX_VARIANT_PATCH_MERGER = 3 };
enum { VITX_POOL_CLS = 0, VITX_POOL_MEAN = 1 };
/* FP32_PARITY: fp32 storage and fp32 FMA everywhere (gates "logits within 1e-3 of the reference").
 * BF16: bf16 GEMM/attention operands on MFMA, fp32 accumulation, statistics and residual stream.
 * BF16X3: the FP32_PARITY data path (fp32 storage, statistics, softmax, GELU, gradients) with every large GEMM evaluated as three bf16 MFMA
 *   products of operands split into bf16 hi + lo parts (~2^-16 per product): the 1e-3 gate at matrix-pipe speed. */
enum { VITX_COMPUTE_FP32_PARITY = 0, VITX_COMPUTE_BF16 = 1, VITX_COMPUTE_BF16X3 = 2 };

/* Mirrors the constructor kwargs 1:1:
 *   ViT(image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool, dim_head, dropout,
 *       emb_dropout)                                   vit_tensorflow/vit.py:107-108
 *   DeepViT(... same ...)                              vit_tensorflow/deepvit.py:113-114
 *   CaiT(..., cls_depth, ..., layer_dropout)           vit_tensorflow/cait.py:156-157 */
typedef struct vitx_config {
  int32_t variant;
  int32_t image_h, image_w;   /* pair(image_size)  vit.py:133 */
  int32_t patch_h, patch_w;   /* pair(patch_size)  vit.py:134 */
  int32_t channels;           /* 3 (the reference's Rearrange infers it from the input) */
  int32_t num_classes, dim, depth, cls_depth, heads, dim_head, mlp_dim;
  int32_t pool;               /* vit.py:139 */
  float dropout, emb_dropout, layer_dropout;
  float ln_eps;               /* Keras LayerNormalization default 1e-3 */
  int32_t compute;
  int32_t max_batch;          /* device buffers are sized once for this batch */
  int32_t device_id;
  /* parallel_vit.ViT(..., num_parallel_branches=2) (parallel_vit.py:119-133): every layer is a sum of this many attention blocks
   * followed by a sum of this many feed-forward blocks, each with its own PreNorm (parallel_vit.py:36-42,104-117).  0 or 1 = the
   * ordinary ViT.  variant must be VITX_VARIANT_VIT. */
  int32_t num_parallel_branches;
  /* VITX_VARIANT_PATCH_MERGER = vit_with_patch_merger.ViT(..., patch_merge_layer=None, patch_merge_num_tokens=8)
   * (vit_with_patch_merger.py:136-183): no cls token, mean pooling, and after layer index
   * (patch_merge_layer > 0 ? patch_merge_layer : depth / 2) - 1 the tokens are merged to patch_merge_num_tokens by PatchMerger
   * (LayerNorm + learned-query attention pooling, vit_with_patch_merger.py:42-55,117,131-132). */
  int32_t patch_merge_layer, patch_merge_num_tokens;
  int32_t reserved[5];
} vitx_config;

typedef struct vitx_engine* vitx_handle;

/* per-kernel-class timing collected between vitx_profile_begin/_end (HIP events on the handle's stream) */
typedef struct vitx_kernel_stat {
  char name[48];
  int64_t launches;
  double total_ms;
  double flops;   /* algorithmic FLOPs issued by these launches */
  double bytes;   /* algorithmic HBM bytes (compulsory reads + writes) */
} vitx_kernel_stat;

/* called from vitx_backward* on the host as soon as every kernel producing grads[offset, offset+count)
 * has been enqueued on the handle's stream (gradient buckets for overlapped all-reduce) */
typedef void (*vitx_grad_ready_fn)(void* user, int64_t offset_elems, int64_t count_elems);

const char* vitx_version(void);
const char* vitx_last_error(void);

/* ---- parameter table (host only, no GPU needed).  Replaces Keras' model.weights / get_weights()
 * ordering (inherited from tf.keras.Model, used by mae.py:36-38).  Order is explicit and documented
 * in DESIGN.md; layouts are Keras': Dense kernel [in,out], bias [out], LN gamma/beta [d]. */
int32_t vitx_param_table_size(const vitx_config* cfg, int64_t* n_tensors, int64_t* n_elems);
int32_t vitx_param_table_entry(const vitx_config* cfg, int64_t index, char* name, int32_t name_cap,
                               int64_t shape[4], int32_t* rank, int64_t* offset_elems);

/* ---- lifetime: ViT.__init__ (vit.py:107-157), DeepViT.__init__ (deepvit.py:113-137),
 * CaiT.__init__ (cait.py:156-178).  Validation errors carry the reference's assertion text. */
int32_t vitx_create(const vitx_config* cfg, vitx_handle* out);
int32_t vitx_destroy(vitx_handle h);
int32_t vitx_get_config(vitx_handle h, vitx_config* out);   /* the (defaulted) config a handle was built with */

/* ---- weights: Keras set_weights/get_weights (flat fp32 blob in table order) */
int32_t vitx_set_params(vitx_handle h, const float* host_blob, int64_t n_elems);
int32_t vitx_get_params(vitx_handle h, float* host_blob, int64_t n_elems);
int32_t vitx_get_grads(vitx_handle h, float* host_blob, int64_t n_elems);
/* device arenas (fp32, table order) -- for optimizers / collectives living outside the library */
int32_t vitx_params_dev(vitx_handle h, float** dev_ptr, int64_t* n_elems);
int32_t vitx_grads_dev(vitx_handle h, float** dev_ptr, int64_t* n_elems);
/* use caller-owned device arenas (e.g. torch tensors) instead of the library's; either may be NULL */
int32_t vitx_bind_arenas(vitx_handle h, float* params_dev, float* grads_dev);
/* tell the engine the fp32 params changed on device (re-derives the bf16 operand copies) */
int32_t vitx_params_changed(vitx_handle h);

/* ---- forward: ViT.call / DeepViT.call / CaiT.call (vit.py:159-177, deepvit.py:139-157,
 * cait.py:180-194).  img NHWC fp32 [b,H,W,C]; H,W may be smaller than the configured image as long
 * as they divide by the patch (pos_embedding is sliced, vit.py:165).  logits [b,num_classes]. */
int32_t vitx_forward(vitx_handle h, const float* img_host, int32_t b, int32_t H, int32_t W,
                     int32_t training, uint64_t seed, float* logits_host);
int32_t vitx_forward_dev(vitx_handle h, const float* img_dev, int32_t b, int32_t H, int32_t W,
                         int32_t training, uint64_t seed, float* logits_dev);

/* ---- backward: the VJP TensorFlow's GradientTape would compute for the forward above
 * (README.md:746-749 is the reference's only mention).  Requires a preceding forward with the same
 * b; fills the gradient arena (overwrites, no accumulation).  dimg may be NULL (TF does not
 * differentiate w.r.t. an un-watched image). */
int32_t vitx_backward(vitx_handle h, const float* dlogits_host, float* dimg_host_or_null);
/* Forward from caller-supplied patch rows [b, np, patch_dim] (fp32) in place of the image + Rearrange: the first layer(s) of T2T-ViT's
 * patch_embedding are a tokenizer pipeline (t2t.py:59-77) whose output feeds the same Dense, cls / position rows, transformer and head
 * (t2t.py:99-121).  A backward after it returns d(patches) [b, np, patch_dim] through the dimg argument. */
/* One shot: the NEXT host-pointer forward entry on this handle (vitx_forward, vitx_forward_distill, or vitx_distill_forward of a
 * wrapper around it) reads its image argument as patch rows [b, np, patch_dim]; H / W are ignored there.  np = 0 clears it. */
int32_t vitx_set_patch_input(vitx_handle h, int32_t np);
int32_t vitx_forward_patches(vitx_handle h, const float* patches_host, int32_t b, int32_t np, int32_t training, uint64_t seed,
                             float* logits_host);
int32_t vitx_forward_patches_dev(vitx_handle h, const float* patches_dev, int32_t b, int32_t np, int32_t training, uint64_t seed,
                                 float* logits_dev_or_null);
int32_t vitx_backward_dev(vitx_handle h, const float* dlogits_dev, float* dimg_dev_or_null);

/* ---- encoder.transformer(tokens, training=training) on arbitrary [b,n,dim] tokens (mae.py:69, simmim.py:116,
 * efficient.py:47, mpp.py:212).  ViT / DeepViT only.  training != 0 applies the model's Dropout layers (vit.py:41,43,64) with
 * masks drawn from `seed`, as vitx_forward does; the VJP below replays them. */
int32_t vitx_transformer_forward(vitx_handle h, const float* tokens_host, int32_t b, int32_t n,
                                 int32_t training, uint64_t seed, float* out_host);
/* VJP of the call above (what GradientTape gives the wrappers that train through encoder.transformer: mae.py:69 + README.md:746-749):
 * d(out) [b,n,dim] -> d(tokens) [b,n,dim] (may be NULL); the gradient arena holds the transformer's parameter gradients, every
 * other entry is zero.  Requires a preceding vitx_transformer_forward (same handle, no full forward in between). */
int32_t vitx_transformer_backward(vitx_handle h, const float* dout_host, float* dtokens_host_or_null);

/* ---- stand-alone patch unfold: Rearrange('b (h p1) (w p2) c -> b (h w) (p1 p2 c)') (vit.py:142,
 * deepvit.py:122, cait.py:164).  Pure index arithmetic: bit-exact. */
int32_t vitx_patch_unfold(const float* img_host, int32_t b, int32_t H, int32_t W, int32_t C,
                          int32_t ph, int32_t pw, float* out_host);

/* ---- efficient.ViT(image_size, patch_size, num_classes, dim, transformer, pool='cls') (efficient.py:12-56): the model is a shell
 * around a transformer object supplied by the caller.  On a ViT handle (normally built with depth = 0; the block parameters of a
 * deeper handle are simply not used) the four entry points below are that shell and its VJP:
 *   embed_forward   patch_embedding + cls token + pos_embedding[:, :n+1] (efficient.py:40-46): img NHWC -> tokens [b, np+1, dim]
 *   head_forward    pooling + mlp_head (efficient.py:49-54): x [b, n, dim] (whatever the transformer returned) -> logits
 *   head_backward   d(logits) -> d(x) [b, n, dim]; fills the mlp_head.* entries of the gradient arena
 *   embed_backward  d(tokens) [b, np+1, dim] -> fills pos_embedding (rows beyond np+1 zero), cls_token, patch_embedding.*;
 *                   optional d(img)
 * Validation error text follows efficient.py:18.  The _dev forms take device pointers and stay asynchronous on the handle's
 * stream (dlogits_dev NULL = the engine's internal dlogits); tokens / x are contiguous fp32. */
int32_t vitx_embed_forward(vitx_handle h, const float* img_host, int32_t b, int32_t H, int32_t W, float* tokens_host);
int32_t vitx_embed_forward_dev(vitx_handle h, const float* img_dev, int32_t b, int32_t H, int32_t W, float* tokens_dev);
int32_t vitx_head_forward(vitx_handle h, const float* x_host, int32_t b, int32_t n, float* logits_host);
int32_t vitx_head_forward_dev(vitx_handle h, const float* x_dev, int32_t b, int32_t n, float* logits_dev_or_null);
int32_t vitx_head_backward(vitx_handle h, const float* dlogits_host, float* dx_host);
int32_t vitx_head_backward_dev(vitx_handle h, const float* dlogits_dev_or_null, float* dx_dev);
int32_t vitx_embed_backward(vitx_handle h, const float* dtokens_host, float* dimg_host_or_null);
int32_t vitx_embed_backward_dev(vitx_handle h, const float* dtokens_dev, float* dimg_dev_or_null);

/* ---- encoder.patch_embedding.layers[1] on its own: the nn.Dense(units=dim) of vit.py:143 as the wrappers borrow it (mae.py:37,52;
 * simmim.py:79,92; mpp.py:200) -- rows of unfolded patches [rows, p1*p2*C] -> [rows, dim]; no cls token, no position embedding.
 * Forward only (the trainable wrappers are vitx_mim_* / vitx_distill_*). */
int32_t vitx_patch_dense_forward(vitx_handle h, const float* patches_host, int32_t rows, float* out_host);

/* ---- T2T tokenizer: tf.image.extract_patches(x, sizes=[1,k,k,1], strides=[1,s,s,1], rates=[1,1,1,1], padding='SAME')
 * (RearrangeUnfoldTransformer.call, t2t.py:39-47) on NHWC x [b,H,W,C] -> [b, ceil(H/s), ceil(W/s), k*k*C], feature order
 * (ki, kj, c), TensorFlow's SAME rule (pad_total = max((out-1)*s + k - in, 0), pad_before = pad_total / 2, zeros outside).
 * Pure index arithmetic: bit-exact.  _backward is its VJP (overlapping windows summed in a fixed order).  Handle-free; the _dev
 * forms enqueue on `hip_stream` (NULL = the default stream). */
int32_t vitx_extract_patches_shape(int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t* oh, int32_t* ow, int32_t* feat);
int32_t vitx_extract_patches(const float* x_host, int32_t b, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, float* out_host);
int32_t vitx_extract_patches_backward(const float* dout_host, int32_t b, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride,
                                      float* dx_host);
int32_t vitx_extract_patches_dev(const float* x_dev, int32_t b, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, float* out_dev,
                                 void* hip_stream);
int32_t vitx_extract_patches_backward_dev(const float* dout_dev, int32_t b, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride,
                                          float* dx_dev, void* hip_stream);

/* ---- loss gradient on device: d/dlogits of mean softmax cross-entropy
 * (tf.keras.losses.categorical_crossentropy(from_logits=True), distill.py:119).  Writes the
 * engine's internal dlogits (used by vitx_backward_dev(h, NULL, ...)) and optionally the mean loss. */
int32_t vitx_ce_loss_grad_dev(vitx_handle h, const int32_t* labels_dev, float inv_global_batch,
                              float* loss_dev_or_null);

/* ---- optimizer step on the device arenas ("next" row f1 of SURVEY.md section 8: the reference has no optimizer or trainer at
 * all; this turns forward+backward into a training step).  Uses the gradients of the last backward (all-reduced or not). */
int32_t vitx_adamw_step(vitx_handle h, float lr, float beta1, float beta2, float eps, float weight_decay);
int32_t vitx_sgd_step(vitx_handle h, float lr, float momentum, float weight_decay);

/* Optimizer state of a handle (AdamW first / second moments, SGD momentum: fp32 in the packed host order of vitx_get_params; the
 * AdamW step count).  A handle owns this state, so a caller that rebuilds its handle (a larger max_batch) carries it across with
 * this pair; have_m / have_v report which buffers exist, NULL host pointers skip the copy. */
int32_t vitx_get_opt_state(vitx_handle h, float* m_host_or_null, float* v_host_or_null, int64_t n_elems, int64_t* step,
                           int32_t* have_m, int32_t* have_v);
int32_t vitx_set_opt_state(vitx_handle h, const float* m_host_or_null, const float* v_host_or_null, int64_t n_elems, int64_t step);

/* ---- streams / sync */
int32_t vitx_set_stream(vitx_handle h, void* hip_stream); /* NULL = the handle's own stream */
int32_t vitx_sync(vitx_handle h);

/* ---- HIP graphs (no reference counterpart: replaces ~300 launches of a step by one when the step is launch-bound, e.g. the
 * README's batch-1 example).  Everything enqueued on the handle through the *_dev entry points between begin and end is recorded
 * instead of run; the captured device pointers (images, labels, logits) are baked in.  Run one eager step of the same geometry
 * first (first-use allocations and the GEMM variant measurements cannot be captured). */
int32_t vitx_graph_capture_begin(vitx_handle h);
int32_t vitx_graph_capture_end(vitx_handle h, void** graph_exec_out);
int32_t vitx_graph_launch(vitx_handle h, void* graph_exec);
int32_t vitx_graph_destroy(void* graph_exec);

/* ---- data parallel (no reference counterpart: batch-sharded replicas, mean-reduced gradients) */
int32_t vitx_set_grad_ready_callback(vitx_handle h, vitx_grad_ready_fn fn, void* user);
int32_t vitx_comm_unique_id(void* out_128_bytes);
int32_t vitx_comm_init(vitx_handle h, int32_t rank, int32_t world, const void* unique_id_128_bytes);
/* Overlapped exchange (SURVEY.md 8(e), Appendix B "grads live in one contiguous arena ... buckets are contiguous slices; side stream + events"):
 * enable != 0 -> from the next backward on, every bucket of the gradient arena is all-reduced (and scaled by 1/world) on the handle's own
 * communication stream the moment the backward pass has enqueued its last producer; vitx_allreduce_grads then only sends what is left and orders the
 * compute stream behind the last bucket.  bucket_bytes: fp32 bytes per bucket (0 = 32 MiB); wire_bf16 != 0: the collective moves bf16 copies
 * (half the xGMI bytes; each rank's addend is rounded once).  Call after vitx_comm_init, between steps. */
int32_t vitx_comm_overlap(vitx_handle h, int32_t enable, int64_t bucket_bytes, int32_t wire_bf16);
/* out4 = {buckets, buckets the last exchange sent from inside the backward pass, elements per bucket, Dense launches that found a collective in flight} */
int32_t vitx_comm_stats(vitx_handle h, int64_t* out4);
/* leave the group: the communicator, the communication stream and the wire buffer are released and the handle is single-rank again (a later
 * vitx_comm_init may join another group).  Call between steps -- an exchange in progress is joined first. */
int32_t vitx_comm_destroy(vitx_handle h);
int32_t vitx_allreduce_grads(vitx_handle h); /* RCCL mean over ranks of the gradient arena (bucketed; finishes an overlapped exchange) */

/* ---- measurement / debugging */
/* Every VITX_* environment variable the library reads, one line per switch: "NAME<TAB>class<TAB>state<TAB>what it does\n" with class in
 * {tuning (same results), path (another validated code path; results within the same oracle gates), diag (timing experiment: may corrupt
 * results)} and state in {unset, set=<value>, ignored (a diag switch in a release build)}.  A release library ignores the diag class; only
 * lib/libvitx_diag.so (build.py --diag) honours it.  Writes at most cap bytes (NUL-terminated); *needed = bytes of the full text + 1. */
int32_t vitx_debug_switches(char* out, int64_t cap, int64_t* needed);
int32_t vitx_profile_begin(vitx_handle h);
int32_t vitx_profile_end(vitx_handle h, vitx_kernel_stat* out, int32_t cap, int32_t* n_out);
int32_t vitx_workspace_bytes(vitx_handle h, int64_t* bytes);
/* copy a saved activation of the last forward to the host as fp32 (bisecting parity failures);
 * `which` in {"embed","x_in","y1","qkv","attn_out","x_mid","y2","hpre","act","x_out","pooled"} */
int32_t vitx_debug_read(vitx_handle h, const char* which, int32_t layer, float* out_host,
                        int64_t cap_elems, int64_t* n_elems);
/* raw GEMM micro-benchmark on the handle's stream: C[M,N] = A[M,K] * B[N,K]^T (bf16 operands,
 * random data), returns the average ms over `iters` launches; `kernel` selects the tile variant */
int32_t vitx_bench_gemm(vitx_handle h, int32_t M, int32_t N, int32_t K, int32_t kernel,
                        int32_t epilogue, int32_t iters, float* avg_ms, float* max_abs_err);
/* full-size self-check of the bf16 MFMA GEMM launches, compared on the device with the k-ordered fp32-FMA kernel on the same bf16 operands
 * (test hook; the Dense layers of vit.py:39,42,59,63 at the benchmarked row count).  kind 0: C[M,N] = A[M,K] B[N,K]^T with fused epilogue
 * `epilogue` (codes of vitx_bench_gemm) on tile variant `kernel`; kind 1: the weight-gradient form dW[M=in, N=out] over K token rows with the
 * engine's split-K rule and slice reduction.  errs2[0] = max |got - want| / (1 + |want|); errs2[1] = the same for the second output
 * (epilogue 2) / the fused column sums (epilogue 4), the number of K slices (kind 1), else -1. */
int32_t vitx_check_gemm(vitx_handle h, int32_t kind, int32_t M, int32_t N, int32_t K, int32_t kernel,
                        int32_t epilogue, float* errs2);

/* ---- masked-image-modelling wrappers around a built encoder ("next" row f2 of SURVEY.md section 8):
 *   MAE(image_size, encoder, decoder_dim, masking_ratio, decoder_depth, decoder_heads, decoder_dim_head)   mae.py:17-45
 *   SimMIM(image_size, encoder, masking_ratio)                                                            simmim.py:68-84
 * The wrapper borrows the encoder handle (to_patch / patch_to_emb / pos_embedding[:, 1:] / .transformer: mae.py:36-38,
 * simmim.py:79-81) and owns its own parameters: MAE {enc_to_dec.kernel/.bias (only when encoder dim != decoder_dim, else the
 * reference's Identity), mask_token, decoder_pos_emb.embeddings, to_pixels.kernel/.bias} plus a decoder Transformer that is an
 * ordinary handle (vitx_mim_decoder: use its "transformer.*" table entries); SimMIM {mask_token, to_pixels.kernel/.bias}.
 * The random masking indices are an input (the reference draws them with tf.random.uniform + argsort / top_k, mae.py:58,
 * simmim.py:108): MAE int32 [b, num_patches] = rand_indices, whose first num_masked columns are the masked patches;
 * SimMIM int32 [b, num_masked] = masked_indices.  num_masked = int(masking_ratio * num_patches). */
/* MPP(image_size, transformer, patch_size, output_channel_bits, channels, max_pixel_val, mask_prob, replace_prob, random_patch_prob,
 *     mean, std)                                                                                              mpp.py:133-218
 * Masked patch prediction: the encoder's own embedding (patch Dense + cls + pos + dropout, mpp.py:200-209), its transformer on all
 * tokens (:212), a Dense to 2^(bits * channels) classes (`to_bits`, :149,213) and a loss over the masked positions (MPPLoss, :90-131).
 * Parameters {to_bits.kernel/.bias, mask_token}.  indices: int32 [b, num_masked] = the top_k draw of get_mask_subset_with_prob (:79-88),
 * num_masked = ceil(mask_prob * num_patches).  Two places where the reference's code does not do what it evidently means are reproduced
 * literally by default (literal_loss = 1) and documented in DESIGN.md: the in-place replacements of mpp.py:185,190 are written into
 * `.numpy()` COPIES and never reach the tensor (the transformer sees the unmasked patches; mask_token receives no gradient), and
 * MPPLoss passes (predictions, labels) to tf.nn.softmax_cross_entropy_with_logits(labels, logits) in swapped order with
 * clip_by_value(target, max_pixel_val, max_pixel_val), so the loss as written is log(2^(bits c)) * mean_i sum_j logits_ij.
 * literal_loss = 0: softmax cross-entropy of the masked positions' logits against the discretised mean colour of their patches. */
enum { VITX_MIM_MAE = 0, VITX_MIM_SIMMIM = 1, VITX_MIM_MPP = 2 };
typedef struct vitx_mim_config {
  int32_t kind;
  int32_t decoder_dim, decoder_depth, decoder_heads, decoder_dim_head; /* MAE only (mae.py:21-25) */
  /* MAE: 1 = the loss exactly as written, tf.reduce_mean(tf.square(pred_pixel_values, masked_patches)) (mae.py:90), where the
   * second positional argument of tf.square is `name`, i.e. mean(pred^2); 0 = the evidently intended mean((pred - masked_patches)^2).
   * SimMIM ignores it (simmim.py:128 is a plain L1). */
  int32_t literal_loss;
  double masking_ratio;            /* MPP: mask_prob */
  int32_t output_channel_bits;     /* MPP (mpp.py:137), default 3 */
  float max_pixel_val;             /* MPP (mpp.py:139), default 1.0 */
  int32_t has_norm;                /* MPP: 1 = un-normalise the target with norm_mean / norm_std (mpp.py:108-109), channels <= 4 */
  float norm_mean[4], norm_std[4];
  int32_t reserved[5];
} vitx_mim_config;
typedef struct vitx_mim* vitx_mim_handle;

int32_t vitx_mim_create(vitx_handle encoder, const vitx_mim_config* cfg, vitx_mim_handle* out);
int32_t vitx_mim_destroy(vitx_mim_handle m);           /* does not destroy the encoder */
vitx_handle vitx_mim_decoder(vitx_mim_handle m);       /* MAE: Transformer(dim=decoder_dim, ..., mlp_dim=4*decoder_dim) mae.py:43; else NULL */
int32_t vitx_mim_param_table_size(vitx_mim_handle m, int64_t* n_tensors, int64_t* n_elems);
int32_t vitx_mim_param_table_entry(vitx_mim_handle m, int64_t index, char* name, int32_t name_cap,
                                   int64_t shape[4], int32_t* rank, int64_t* offset_elems);
int32_t vitx_mim_set_params(vitx_mim_handle m, const float* host_blob, int64_t n_elems);
int32_t vitx_mim_get_params(vitx_mim_handle m, float* host_blob, int64_t n_elems);
int32_t vitx_mim_get_grads(vitx_mim_handle m, float* host_blob, int64_t n_elems);
int32_t vitx_mim_params_dev(vitx_mim_handle m, float** params_dev, float** grads_dev, int64_t* n_arena_elems);
int32_t vitx_mim_params_changed(vitx_mim_handle m);   /* the wrapper's fp32 params changed on device (like vitx_params_changed) */
int32_t vitx_mim_num_masked(vitx_mim_handle m, int32_t H, int32_t W, int32_t* num_patches, int32_t* num_masked);
/* MAE.call (mae.py:47-92) / SimMIM.call (simmim.py:86-130): returns the reconstruction loss.  The host variant validates the
 * indices (range, distinct per image: simmim.py:31) and synchronises; the _dev variant trusts them and stays asynchronous. */
int32_t vitx_mim_forward(vitx_mim_handle m, const float* img_host, int32_t b, int32_t H, int32_t W,
                         const int32_t* indices_host, int32_t training, uint64_t seed, float* loss_host);
int32_t vitx_mim_forward_dev(vitx_mim_handle m, const float* img_dev, int32_t b, int32_t H, int32_t W,
                             const int32_t* indices_dev, int32_t training, uint64_t seed, float* loss_dev_or_null);
/* VJP of the forward above for d(loss) = 1 (README.md:746-749 style training): wrapper gradients -> vitx_mim_get_grads,
 * encoder gradients -> the encoder's arena (transformer blocks, patch_embedding.*, pos_embedding rows 1..num_patches; all other
 * entries zero), decoder gradients -> the decoder handle's arena.  Unlike the reference -- whose `.numpy()` indexing
 * (mae.py:62, simmim.py:119) silently detaches everything upstream of it from the tape -- the gather is differentiable. */
int32_t vitx_mim_backward(vitx_mim_handle m);
/* copy a tensor of the last forward to the host: "pred", "target", "patches", "encoded", "decoded" (MAE) */
int32_t vitx_mim_read(vitx_mim_handle m, const char* which, float* out_host, int64_t cap_elems, int64_t* n_elems);

/* ---- knowledge distillation ("next" row f3 of SURVEY.md section 8).
 * DistillableViT.call(img, distill_token) (DistillMixin, distill.py:16-44): the token [dim] is appended after the position
 * embedding, attended with the other tokens, split off before pooling; returns logits [b,num_classes] and the per-image
 * distillation tokens [b,dim].  backward: cotangents of both outputs -> gradient arena, d(token) [dim], optional d(img).
 * Works on any ViT / DeepViT handle (DistillableViT adds no parameters of its own, distill.py:46-57). */
int32_t vitx_forward_distill(vitx_handle h, const float* img_host, int32_t b, int32_t H, int32_t W, int32_t training, uint64_t seed,
                             const float* distill_token_host, float* logits_host, float* distill_tokens_host);
int32_t vitx_backward_distill(vitx_handle h, const float* dlogits_host, const float* d_distill_tokens_host_or_null,
                              float* d_distill_token_host_or_null, float* dimg_host_or_null);

/* DistillWrapper(teacher, student, temperature, alpha, hard) (distill.py:87-134).  The teacher is any model: its logits are an
 * input (tf.stop_gradient, distill.py:114).  Own parameters: distillation_token [1,1,dim], distill_mlp.norm.{gamma,beta},
 * distill_mlp.{kernel,bias} (distill.py:101-106).  loss [b] = CE(labels, student_logits) * (1 - alpha) + distill_loss * alpha,
 * with labels [b,num_classes] one-hot or soft (categorical_crossentropy(from_logits=True), distill.py:119). */
typedef struct vitx_distill_config {
  float temperature, alpha; /* distill.py:88 defaults 1.0, 0.5 */
  int32_t hard;             /* 0: temperature-scaled KL to the teacher's softmax (distill.py:121-129); 1: cross-entropy against
                             * the teacher's argmax (distill.py:130-132: as written it passes rank-1 integer labels to
                             * categorical_crossentropy, which TensorFlow rejects; the engine computes the sparse form) */
  /* soft mode, 1 = exactly as written: keras.losses.KLDivergence is handed LOG-probabilities as y_pred (distill.py:122-124) and
   * clips them to [1e-7, 1], so the term is sum(y log(y / 1e-7)), constant in the student (zero gradient);
   * 0 = the evident intent KL(softmax(teacher / T) || softmax(distill_logits / T)) */
  int32_t literal_loss;
  int32_t reserved[8];
} vitx_distill_config;
typedef struct vitx_distill* vitx_distill_handle;

int32_t vitx_distill_create(vitx_handle student, const vitx_distill_config* cfg, vitx_distill_handle* out);
int32_t vitx_distill_destroy(vitx_distill_handle m);   /* does not destroy the student */
int32_t vitx_distill_param_table_size(vitx_distill_handle m, int64_t* n_tensors, int64_t* n_elems);
int32_t vitx_distill_param_table_entry(vitx_distill_handle m, int64_t index, char* name, int32_t name_cap,
                                       int64_t shape[4], int32_t* rank, int64_t* offset_elems);
int32_t vitx_distill_set_params(vitx_distill_handle m, const float* host_blob, int64_t n_elems);
int32_t vitx_distill_get_params(vitx_distill_handle m, float* host_blob, int64_t n_elems);
int32_t vitx_distill_get_grads(vitx_distill_handle m, float* host_blob, int64_t n_elems);
/* DistillWrapper.call((img, labels), temperature, alpha, training) (distill.py:107-134); temperature <= 0 / alpha < 0 take the
 * constructor's values (distill.py:110-111).  loss: [b]. */
int32_t vitx_distill_forward(vitx_distill_handle m, const float* img_host, const float* labels_host, const float* teacher_logits_host,
                             int32_t b, int32_t H, int32_t W, int32_t training, uint64_t seed, float temperature, float alpha,
                             float* loss_host);
int32_t vitx_distill_forward_dev(vitx_distill_handle m, const float* img_dev, const float* labels_dev, const float* teacher_logits_dev,
                                 int32_t b, int32_t H, int32_t W, int32_t training, uint64_t seed, float temperature, float alpha,
                                 float* loss_dev_or_null);
/* VJP for the cotangent dloss [b] (NULL = ones: tape.gradient of a non-scalar target differentiates its sum): wrapper gradients
 * -> vitx_distill_get_grads, student gradients -> the student's arena. */
int32_t vitx_distill_backward(vitx_distill_handle m, const float* dloss_host_or_null);
/* the same, also returning d(loss)/d(input) of the student's forward: d(img) [b,H,W,C], or d(patches) [b,np,patch_dim] when that forward
 * took patch rows (vitx_set_patch_input: a T2T-ViT student, whose tokenizer in front of the handle continues the chain) */
int32_t vitx_distill_backward_input(vitx_distill_handle m, const float* dloss_host_or_null, float* dinput_host);
/* "student_logits", "distill_logits" [b,num_classes], "distill_tokens" [b,dim] of the last forward */
int32_t vitx_distill_read(vitx_distill_handle m, const char* which, float* out_host, int64_t cap_elems, int64_t* n_elems);

#ifdef __cplusplus
}
#endif
#endif /* VITX_H_ */
