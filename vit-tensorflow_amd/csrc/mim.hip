// Masked-image-modelling wrappers around an encoder handle: MAE (mae.py:17-92) and SimMIM (simmim.py:68-130).
// Both take a built ViT, reuse the first two layers of its patch_embedding and rows 1.. of its pos_embedding, run
// encoder.transformer on a token subset / on mask-substituted tokens, and regress pixel values of the masked patches.
// Here the whole step stays on the device: launch sequences over the encoder engine (patch tokens, transformer), a second engine
// for MAE's decoder Transformer, the index kernels of mim_ops.hip and, in parity mode, the exact-fp32 GEMM for the wrappers' own small Dense
// layers (enc_to_dec, to_pixels; bf16 mode: the encoder engine's MFMA GEMMs on bf16 copies of their operands).  The backward is the true VJP of that forward; the reference cuts its gradient tape wherever it
// indexes through `.numpy()` (mae.py:62, simmim.py:119) -- see DESIGN.md for what that means for parity.
#include <cstring>
#include <algorithm>

#include "engine.h"

int capi_fail(int code, const std::string& msg);   // capi.hip (thread-local last-error text)

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      err = std::string(#x) + ": " + hipGetErrorString(e_);                                         \
      return VITX_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)

struct vitx_mim {
  vitx_mim_config cfg{};
  vitx_engine* enc = nullptr;
  vitx_engine* dec = nullptr;        // MAE: Transformer(dim=decoder_dim, ..., mlp_dim=4*decoder_dim) (mae.py:43), owned
  std::vector<ParamDesc> table;
  int64_t n_params = 0, n_arena = 0;
  float *params = nullptr, *grads = nullptr;
  int64_t w_ed = -1, b_ed = -1, mask_tok = -1, dpos = -1, w_px = -1, b_px = -1;   // arena offsets
  int np_max = 0, d = 0, dd = 0, pd = 0, B = 0;
  bool mae = false, project = false;  // project: enc_to_dec is a Dense (encoder_dim != decoder_dim), else Identity (mae.py:41)
  bool mpp = false;                   // MPP (mpp.py:133): px = to_bits, Dense(dim -> 2^(bits * channels))
  int po = 0;                         // output width of the wrapper's last Dense: patch_dim (to_pixels) or 2^(bits * channels) (to_bits)
  float *tok1 = nullptr, *enc1 = nullptr;   // MPP: [B, np + 1, d] embedded tokens / transformer output (cls row included)
  int32_t* labels = nullptr;          // MPP, literal_loss = 0: class of each masked patch
  std::vector<void*> allocs;
  float *img = nullptr, *patches = nullptr, *tok = nullptr, *sel = nullptr, *enc_out = nullptr, *proj = nullptr, *dec_in = nullptr,
        *dec_out = nullptr, *rows_m = nullptr, *pred = nullptr, *target = nullptr, *dpred = nullptr;
  float *g_rows_m = nullptr, *g_a = nullptr, *g_b = nullptr, *g_c = nullptr;   // backward scratch ([B*np, max(d,dd)] each)
  float *ws = nullptr, *loss = nullptr;
  int32_t *idx = nullptr, *inv = nullptr;
  // bf16 mode: the wrapper's Dense layers run on the encoder engine's MFMA GEMMs (row-padded bf16 copies of their operands)
  bool mfma = false, w_dirty = true;
  Dense px, ed;
  void *x_px_T = nullptr, *x_ed_T = nullptr, *dy_px_T = nullptr, *dy_ed_T = nullptr;   // one buffer per (layer, operand): each keeps its own zero tail
  int64_t t_rows = 0, glue_geom = -1;
  bool have_fwd = false;
  int b = 0, np = 0, nm = 0;
  float mpp_edrop = 0.f; uint64_t mpp_seed = 0;   // embedding-dropout mask of the last MPP forward
};

namespace {

// patches the wrapper masks at an image of np patches: int(masking_ratio * np) (mae.py:57, simmim.py:106); MPP: ceil(mask_prob * np) (mpp.py:80)
int mim_num_masked(const vitx_mim* m, int np) {
  if (m->mpp) return std::min(np, (int)std::ceil(m->cfg.masking_ratio * (double)np));
  return (int)(m->cfg.masking_ratio * (double)np);
}

int64_t add_param(vitx_mim* m, const std::string& name, std::vector<int64_t> shape) {
  ParamDesc p;
  p.name = name; p.shape = shape; p.count = 1;
  for (int64_t s : shape) p.count *= s;
  p.offset = m->n_params; p.aoff = m->n_arena;
  m->n_params += p.count;
  m->n_arena += round_up(p.count, 4);
  m->table.push_back(p);
  return p.aoff;
}

// ONE ordering rule for every parameter list of this library: a model's OWN variables first, then its sub-layers in
// the attribute order of the reference constructor, kernel before bias -- what Keras 2's Layer.trainable_weights does (`self._trainable_weights +
// children's`), as far as it can be stated without TensorFlow here (oracle/gen_ref_fixtures.py --real-tf prints the comparison where it exists).
// Shapes are the reference variables' own: MPP's mask token is [1, 1, c p^2] (mpp.py:159), MAE's / SimMIM's are vectors (mae.py:42, simmim.py:83).
void build_mim_table(vitx_mim* m) {
  m->table.clear(); m->n_params = m->n_arena = 0;
  if (m->mpp) {   // mpp.py:159 (variable), mpp.py:149 (to_bits)
    m->mask_tok = add_param(m, "mask_token", {1, 1, m->pd});
    m->w_px = add_param(m, "to_bits.kernel", {m->d, m->po});
    m->b_px = add_param(m, "to_bits.bias", {m->po});
  } else if (m->mae) {   // mae.py:42 (variable), then mae.py:41,44,45
    m->mask_tok = add_param(m, "mask_token", {m->dd});
    if (m->project) { m->w_ed = add_param(m, "enc_to_dec.kernel", {m->d, m->dd}); m->b_ed = add_param(m, "enc_to_dec.bias", {m->dd}); }
    // num_patches is read off pos_embedding.shape[-2] (mae.py:37), i.e. it counts the cls row: the table has np + 1 rows
    m->dpos = add_param(m, "decoder_pos_emb.embeddings", {m->np_max + 1, m->dd});
    m->w_px = add_param(m, "to_pixels.kernel", {m->dd, m->pd});
    m->b_px = add_param(m, "to_pixels.bias", {m->pd});
  } else {
    m->mask_tok = add_param(m, "mask_token", {m->d});
    m->w_px = add_param(m, "to_pixels.kernel", {m->d, m->pd});
    m->b_px = add_param(m, "to_pixels.bias", {m->pd});
  }
}

int mim_alloc(vitx_mim* m, void** p, size_t bytes, std::string& err) {
  bytes = (size_t)round_up((int64_t)std::max<size_t>(bytes, 16), 256);
  HIPCHK(hipMalloc(p, bytes));
  HIPCHK(hipMemsetAsync(*p, 0, bytes, m->enc->stream));
  m->allocs.push_back(*p);
  return VITX_OK;
}
#define MALLOC(ptr, bytes)                                         \
  do {                                                             \
    int rc_ = mim_alloc(m, (void**)&(ptr), (size_t)(bytes), err);  \
    if (rc_ != VITX_OK) return rc_;                                \
  } while (0)

bool aligned16(std::initializer_list<const void*> ps) {
  for (const void* p : ps) if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

// Keras Dense on fp32 rows: y = x @ W[in,out] + bias (mae.py:41,45,70,86; simmim.py:84,122), exact fp32 FMA chain
void lin_fwd(const float* x, int rows, int in, const float* W, const float* bias, int out, float* y, hipStream_t s) {
  GenericGemmArgs g;
  g.A = x; g.B = W; g.M = rows; g.N = out; g.K = in; g.sam = in; g.sak = 1; g.sbk = out; g.sbn = 1;
  EpiParams ep;
  ep.out = y; ep.ldo = out; ep.M = rows; ep.N = out; ep.bias = bias;
  ep.vec_ok = (out % 4 == 0) && aligned16({y, bias});
  launch_gemm_generic(g, ep, EPI_STORE_F32, 0, 0, 0, s);
}
// VJP: dx = dy @ W^T (optional), dW = x^T @ dy, db = column sums of dy (optional)
void lin_bwd(const float* x, const float* W, const float* dy, int rows, int in, int out, float* dx, float* dW, float* db, float* ws, hipStream_t s) {
  if (dx) {
    GenericGemmArgs g;
    g.A = dy; g.B = W; g.M = rows; g.N = in; g.K = out; g.sam = out; g.sak = 1; g.sbk = 1; g.sbn = out;
    EpiParams ep;
    ep.out = dx; ep.ldo = in; ep.M = rows; ep.N = in;
    ep.vec_ok = (in % 4 == 0) && aligned16({dx});
    launch_gemm_generic(g, ep, EPI_STORE_F32, 0, 0, 0, s);
  }
  {
    GenericGemmArgs g;
    g.A = x; g.B = dy; g.M = in; g.N = out; g.K = rows; g.sam = 1; g.sak = in; g.sbk = out; g.sbn = 1;
    EpiParams ep;
    ep.out = dW; ep.ldo = out; ep.M = in; ep.N = out;
    ep.vec_ok = (out % 4 == 0) && aligned16({dW});
    launch_gemm_generic(g, ep, EPI_STORE_F32, 0, 0, 0, s);
  }
  if (db) launch_colsum(dy, 0, out, rows, out, ws, db, s);
}

// bf16 mode: rows >= `rows` of the bf16 operand copies are K padding of the weight-gradient GEMMs and must stay zero
int glue_prepare(vitx_mim* m, int b, int np, std::string& err) {
  if (!m->mfma) return VITX_OK;
  hipStream_t s = m->enc->stream;
  const int64_t geom = ((int64_t)b << 32) | (uint32_t)np;
  if (m->glue_geom >= 0 && m->glue_geom != geom) {
    const int64_t dm = std::max(m->d, m->dd);
    HIPCHK(hipMemsetAsync(m->x_px_T, 0, (size_t)m->t_rows * dm * 2, s));
    HIPCHK(hipMemsetAsync(m->x_ed_T, 0, (size_t)m->t_rows * dm * 2, s));
    HIPCHK(hipMemsetAsync(m->dy_px_T, 0, (size_t)m->t_rows * m->po * 2, s));
    HIPCHK(hipMemsetAsync(m->dy_ed_T, 0, (size_t)m->t_rows * dm * 2, s));
  }
  m->glue_geom = geom;
  if (m->w_dirty) {
    engine_ext_dense_refresh(m->enc, m->px);
    if (m->project) engine_ext_dense_refresh(m->enc, m->ed);
    m->w_dirty = false;
  }
  return VITX_OK;
}
// y = x @ W + bias through whichever GEMM the compute mode prescribes; keeps the bf16 copy of x for the backward
void glue_fwd(vitx_mim* m, const Dense& w, void* xT, const float* x, int rows, int in, const float* W, const float* bias, int out, float* y) {
  hipStream_t s = m->enc->stream;
  if (m->mfma) {
    launch_convert(x, in, xT, 1, in, rows, in, in, s);
    engine_ext_dense_fwd(m->enc, xT, in, rows, w, y);
  } else {
    lin_fwd(x, rows, in, W, bias, out, y, s);
  }
}
void glue_bwd(vitx_mim* m, const Dense& w, const void* xT, void* dyT, const float* x, const float* W, const float* dy, int rows, int in, int out,
              float* dx, float* dW, float* db) {
  hipStream_t s = m->enc->stream;
  if (m->mfma) {
    launch_convert(dy, out, dyT, 1, out, rows, out, out, s);
    engine_ext_dense_bwd(m->enc, xT, in, dyT, out, dy, rows, w, dx);
    if (db) launch_colsum(dy, 0, out, rows, out, m->ws, db, s);
  } else {
    lin_bwd(x, W, dy, rows, in, out, dx, dW, db, m->ws, s);
  }
}

int mim_create(vitx_engine* enc, const vitx_mim_config& cfg, vitx_mim** out, std::string& err) {
  if (!(cfg.masking_ratio > 0.0 && cfg.masking_ratio < 1.0)) { err = "masking ratio must be kept between 0 and 1"; return VITX_ERR_INVALID; }   // mae.py:28, simmim.py:71
  if (cfg.kind != VITX_MIM_MAE && cfg.kind != VITX_MIM_SIMMIM && cfg.kind != VITX_MIM_MPP) { err = "unknown wrapper kind"; return VITX_ERR_INVALID; }
  if (enc->cfg.variant == VITX_VARIANT_CAIT || enc->cfg.variant == VITX_VARIANT_PATCH_MERGER) { err = "MAE / SimMIM / MPP need an encoder with pos_embedding[:, 1:] and .transformer (ViT / DeepViT)"; return VITX_ERR_UNSUPPORTED; }
  if (cfg.kind == VITX_MIM_MPP) {
    const int bits = cfg.output_channel_bits > 0 ? cfg.output_channel_bits : 3;
    if (bits * enc->cfg.channels > 14) { err = "output_channel_bits * channels must be <= 14 (2^14 classes)"; return VITX_ERR_UNSUPPORTED; }
    if (enc->cfg.patch_h != enc->cfg.patch_w) { err = "MPP needs square patches (mpp.py:113: p1 = p2 = patch_size)"; return VITX_ERR_UNSUPPORTED; }
    if (cfg.has_norm && enc->cfg.channels > 4) { err = "mean / std: at most 4 channels"; return VITX_ERR_UNSUPPORTED; }
    if (enc->cfg.num_parallel_branches > 1) { err = "MPP: not for parallel_vit encoders"; return VITX_ERR_UNSUPPORTED; }
  }
  vitx_mim* m = new vitx_mim();
  m->cfg = cfg; m->enc = enc;
  m->mae = cfg.kind == VITX_MIM_MAE;
  m->mpp = cfg.kind == VITX_MIM_MPP;
  if (m->mpp) {
    if (m->cfg.output_channel_bits <= 0) m->cfg.output_channel_bits = 3;       // mpp.py:137
    if (!(m->cfg.max_pixel_val > 0.f)) m->cfg.max_pixel_val = 1.0f;            // mpp.py:139
  }
  m->np_max = enc->np_max; m->d = enc->cfg.dim; m->pd = enc->pd; m->B = enc->cfg.max_batch;
  m->po = m->mpp ? (1 << (m->cfg.output_channel_bits * enc->cfg.channels)) : m->pd;
  m->dd = m->mae ? cfg.decoder_dim : m->d;
  if (m->mae && (cfg.decoder_dim <= 0 || cfg.decoder_depth < 0 || cfg.decoder_heads <= 0 || cfg.decoder_dim_head <= 0)) {
    delete m; err = "decoder_dim, decoder_heads, decoder_dim_head must be positive"; return VITX_ERR_INVALID;
  }
  m->project = m->mae && m->d != m->dd;
  build_mim_table(m);
  if (m->mae) {
    vitx_config dc{};
    dc.variant = VITX_VARIANT_VIT;                      // mae.py:7,43: vit.Transformer whatever the encoder class
    dc.image_h = enc->cfg.image_h; dc.image_w = enc->cfg.image_w; dc.patch_h = enc->cfg.patch_h; dc.patch_w = enc->cfg.patch_w;
    dc.channels = enc->cfg.channels; dc.num_classes = 1; dc.dim = cfg.decoder_dim; dc.depth = cfg.decoder_depth; dc.heads = cfg.decoder_heads;
    dc.dim_head = cfg.decoder_dim_head; dc.mlp_dim = cfg.decoder_dim * 4; dc.pool = VITX_POOL_CLS;
    dc.ln_eps = enc->cfg.ln_eps; dc.compute = enc->cfg.compute; dc.max_batch = enc->cfg.max_batch; dc.device_id = enc->cfg.device_id;
    int rc = engine_create(dc, &m->dec, err);
    if (rc != VITX_OK) { delete m; err = "decoder: " + err; return rc; }
  }
  const int64_t R = (int64_t)m->B * m->np_max, dm = std::max(m->d, m->dd);
  MALLOC(m->params, (size_t)m->n_arena * 4);
  MALLOC(m->grads, (size_t)m->n_arena * 4);
  MALLOC(m->img, (size_t)m->B * enc->cfg.image_h * enc->cfg.image_w * enc->cfg.channels * 4);
  MALLOC(m->patches, (size_t)R * m->pd * 4);
  MALLOC(m->tok, (size_t)R * m->d * 4);
  MALLOC(m->enc_out, (size_t)R * m->d * 4);
  MALLOC(m->rows_m, (size_t)R * dm * 4);
  MALLOC(m->pred, (size_t)R * m->po * 4);
  MALLOC(m->target, (size_t)R * m->pd * 4);
  MALLOC(m->dpred, (size_t)R * m->po * 4);
  if (m->mpp) {
    MALLOC(m->tok1, (size_t)m->B * (m->np_max + 1) * m->d * 4);
    MALLOC(m->enc1, (size_t)m->B * (m->np_max + 1) * m->d * 4);
    MALLOC(m->labels, (size_t)R * 4);
  }
  if (m->mae) {
    MALLOC(m->sel, (size_t)R * m->d * 4);
    MALLOC(m->proj, (size_t)R * m->dd * 4);
    MALLOC(m->dec_in, (size_t)R * m->dd * 4);
    MALLOC(m->dec_out, (size_t)R * m->dd * 4);
  }
  MALLOC(m->g_rows_m, (size_t)R * dm * 4);
  MALLOC(m->g_a, (size_t)R * dm * 4);
  MALLOC(m->g_b, (size_t)R * dm * 4);
  MALLOC(m->g_c, (size_t)R * dm * 4);
  const int64_t ws_elems = std::max<int64_t>({colsum_ws_elems((int)std::max<int64_t>({(int64_t)m->pd, (int64_t)m->po, dm})), recon_loss_ws_elems(R * std::max(m->pd, m->po)), (int64_t)m->B * dm}) + 64;
  MALLOC(m->ws, (size_t)ws_elems * 4);
  MALLOC(m->loss, 256);
  MALLOC(m->idx, (size_t)R * 4);
  MALLOC(m->inv, (size_t)R * 4);
  m->mfma = enc->bf16 && !enc->force_generic_gemm && m->d % 64 == 0 && m->dd % 64 == 0 && m->po % 64 == 0;
  if (m->mfma) {
    int rc;
    const int in_px = m->mae ? m->dd : m->d;
    if ((rc = engine_ext_dense_init(enc, m->px, in_px, m->po, m->params + m->w_px, m->params + m->b_px, m->grads + m->w_px, m->grads + m->b_px, err)) != VITX_OK) return rc;
    if (m->project &&
        (rc = engine_ext_dense_init(enc, m->ed, m->d, m->dd, m->params + m->w_ed, m->params + m->b_ed, m->grads + m->w_ed, m->grads + m->b_ed, err)) != VITX_OK) return rc;
    m->t_rows = round_up(R, 256) + 384;   // GEMM tiles may read up to 319 rows past M
    MALLOC(m->x_px_T, (size_t)m->t_rows * dm * 2);
    MALLOC(m->x_ed_T, (size_t)m->t_rows * dm * 2);
    MALLOC(m->dy_px_T, (size_t)m->t_rows * m->po * 2);
    MALLOC(m->dy_ed_T, (size_t)m->t_rows * dm * 2);
  }
  HIPCHK(hipStreamSynchronize(enc->stream));
  *out = m;
  return VITX_OK;
}

void mim_destroy(vitx_mim* m) {
  if (!m) return;
  (void)hipDeviceSynchronize();
  for (void* p : m->allocs) (void)hipFree(p);
  if (m->dec) engine_destroy(m->dec);
  delete m;
}

// mae.py:47-92 / simmim.py:86-130 on device.  idx_dev: MAE int32 [b, np] = rand_indices (first num_masked columns are the masked
// patches, mae.py:58-59); SimMIM int32 [b, num_masked] = masked_indices (simmim.py:108).
int mim_forward(vitx_mim* m, const float* img_dev, int b, int H, int W, const int32_t* idx_dev, int training, uint64_t seed, std::string& err) {
  vitx_engine* e = m->enc;
  hipStream_t s = e->stream;
  if (m->dec) m->dec->stream = s;
  const vitx_config& c = e->cfg;
  if (b <= 0 || b > c.max_batch) { err = "batch must be in [1, max_batch]"; return VITX_ERR_INVALID; }
  if (H <= 0 || W <= 0 || H > c.image_h || W > c.image_w || H % c.patch_h || W % c.patch_w) {
    err = "Image dimensions must be divisible by the patch size."; return VITX_ERR_INVALID;
  }
  const int np = (H / c.patch_h) * (W / c.patch_w), d = m->d, dd = m->dd, pd = m->pd;
  const int nm = mim_num_masked(m, np);
  const int nu = np - nm;
  if (nm <= 0 || (nu <= 0 && !m->mpp)) { err = "masking ratio leaves no masked (or no visible) patch at this image size"; return VITX_ERR_UNSUPPORTED; }
  m->have_fwd = false;
  int rc;
  if ((rc = glue_prepare(m, b, np, err)) != VITX_OK) return rc;
  if (m->mpp) {
    // MPP.call (mpp.py:166-218).  The replacements of mpp.py:177-190 are assignments into `.numpy()` copies: masked_input stays the patches.
    const float* Pm = m->params;
    const int nb = m->po, ntok = np + 1;
    if ((rc = engine_embed_forward(e, img_dev, b, H, W, m->tok1, err)) != VITX_OK) return rc;                 // mpp.py:200-206: patch Dense, cls, pos
    const float edrop = training ? c.emb_dropout : 0.f;
    if (edrop > 0.f) launch_dropout(m->tok1, 0, (int64_t)b * ntok * d, edrop, seed, 0u, s);                     // transformer.dropout (mpp.py:209) = nn.Dropout(emb_dropout), vit.py:148
    if ((rc = engine_transformer_forward(e, m->tok1, b, ntok, training, seed, m->enc1, err)) != VITX_OK) return rc;          // mpp.py:212
    // to_bits is applied row by row and the loss reads the masked positions only (mpp.py:213-216, :125): those rows are gathered first
    launch_gather_rows(m->enc1 + d, (int64_t)ntok * d, idx_dev, nm, 0, b, nm, d, m->rows_m, s);
    glue_fwd(m, m->px, m->x_px_T, m->rows_m, b * nm, d, Pm + m->w_px, Pm + m->b_px, nb, m->pred);
    if (m->cfg.literal_loss) {
      // tf.nn.softmax_cross_entropy_with_logits(labels = predictions, logits = label ids [n, 1] broadcast): log_softmax of nb equal logits is
      // -log(nb), so the loss is log(nb) * mean_i sum_j pred_ij whatever the labels are (mpp.py:125-126)
      launch_recon_loss(m->pred, nullptr, (int64_t)b * nm * nb, 2, (float)(std::log((double)nb) / ((double)b * nm)), m->dpred, m->ws, m->loss, s);
    } else {
      launch_mpp_labels(img_dev, b, H, W, c.channels, c.patch_h, m->cfg.output_channel_bits, m->cfg.max_pixel_val, m->cfg.has_norm, m->cfg.norm_mean,
                        m->cfg.norm_std, idx_dev, nm, m->labels, s);
      launch_ce_grad(m->pred, nb, m->labels, b * nm, nb, (float)(1.0 / ((double)b * nm)), m->dpred, m->target, s);   // per-row losses (already x 1/rows) ...
      launch_sum_rows(m->target, b * nm, 1, m->loss, s);                                                                // ... summed in row order
    }
    if (idx_dev != m->idx) HIPCHK(hipMemcpyAsync(m->idx, idx_dev, (size_t)b * nm * 4, hipMemcpyDeviceToDevice, s));
    m->have_fwd = true; m->b = b; m->np = np; m->nm = nm;
    m->mpp_edrop = edrop; m->mpp_seed = seed;
    return VITX_OK;
  }
  if ((rc = engine_patch_tokens_forward(e, img_dev, b, H, W, m->tok, m->patches, err)) != VITX_OK) return rc;   // mae.py:49-55 / simmim.py:88-100
  const float* P = m->params;
  float scale;
  int kind;
  const float* target = nullptr;
  if (m->mae) {
    launch_index_inverse(idx_dev, np, b, 0, np, np, m->inv, s);
    launch_gather_rows(m->tok, (int64_t)np * d, idx_dev, np, nm, b, nu, d, m->sel, s);                      // mae.py:62
    if ((rc = engine_transformer_forward(e, m->sel, b, nu, training, seed, m->enc_out, err)) != VITX_OK) return rc;          // mae.py:69
    const float* proj = m->enc_out;
    if (m->project) { glue_fwd(m, m->ed, m->x_ed_T, m->enc_out, b * nu, d, P + m->w_ed, P + m->b_ed, dd, m->proj); proj = m->proj; }   // mae.py:72
    launch_mae_assemble(proj, P + m->mask_tok, P + m->dpos, idx_dev, b, np, nm, dd, m->dec_in, s);          // mae.py:75-82
    if ((rc = engine_transformer_forward(m->dec, m->dec_in, b, np, training, seed + 0x9e3779b97f4a7c15ull, m->dec_out, err)) != VITX_OK) return rc;  // mae.py:83
    HIPCHK(hipMemcpy2DAsync(m->rows_m, (size_t)nm * dd * 4, m->dec_out, (size_t)np * dd * 4, (size_t)nm * dd * 4, b, hipMemcpyDeviceToDevice, s));   // mae.py:86
    glue_fwd(m, m->px, m->x_px_T, m->rows_m, b * nm, dd, P + m->w_px, P + m->b_px, pd, m->pred);            // mae.py:87
    if (!m->cfg.literal_loss) {
      launch_gather_rows(m->patches, (int64_t)np * pd, idx_dev, np, 0, b, nm, pd, m->target, s);           // mae.py:65
      target = m->target;
    }
    kind = 0;
    scale = (float)(1.0 / ((double)b * nm * pd));                                                           // mae.py:90
  } else {
    launch_index_inverse(idx_dev, nm, b, 0, nm, np, m->inv, s);                                             // simmim.py:109-110 (the bool mask)
    launch_simmim_select(m->tok, m->inv, P + m->mask_tok, e->params + e->pos + d, b, np, d, s);             // simmim.py:102-113
    if ((rc = engine_transformer_forward(e, m->tok, b, np, training, seed, m->enc_out, err)) != VITX_OK) return rc;          // simmim.py:116
    launch_gather_rows(m->enc_out, (int64_t)np * d, idx_dev, nm, 0, b, nm, d, m->rows_m, s);                // simmim.py:119
    glue_fwd(m, m->px, m->x_px_T, m->rows_m, b * nm, d, P + m->w_px, P + m->b_px, pd, m->pred);             // simmim.py:122
    launch_gather_rows(m->patches, (int64_t)np * pd, idx_dev, nm, 0, b, nm, pd, m->target, s);              // simmim.py:125
    target = m->target;
    kind = 1;
    scale = (float)(1.0 / ((double)b * nm * pd * nm));                                                      // simmim.py:128
  }
  launch_recon_loss(m->pred, target, (int64_t)b * nm * pd, kind, scale, m->dpred, m->ws, m->loss, s);
  if (idx_dev != m->idx) HIPCHK(hipMemcpyAsync(m->idx, idx_dev, (size_t)b * (m->mae ? np : nm) * 4, hipMemcpyDeviceToDevice, s));
  m->have_fwd = true; m->b = b; m->np = np; m->nm = nm;
  return VITX_OK;
}

// d(loss) = 1: wrapper gradients -> m->grads, encoder gradients -> the encoder's arena (transformer blocks, patch_embedding,
// pos_embedding rows 1..np; everything else zero), decoder gradients -> the decoder handle's arena.
int mim_backward(vitx_mim* m, std::string& err) {
  if (!m->have_fwd) { err = "backward requires a preceding forward"; return VITX_ERR_STATE; }
  vitx_engine* e = m->enc;
  hipStream_t s = e->stream;
  if (m->dec) m->dec->stream = s;
  const int b = m->b, np = m->np, nm = m->nm, nu = np - nm, d = m->d, dd = m->dd, pd = m->pd;
  const float* P = m->params;
  float* G = m->grads;
  int rc;
  launch_fill_zero(G, m->n_arena * 4, s);
  if (m->mpp) {
    const int nb = m->po, ntok = np + 1;
    glue_bwd(m, m->px, m->x_px_T, m->dy_px_T, m->rows_m, P + m->w_px, m->dpred, b * nm, d, nb, m->g_rows_m, G + m->w_px, G + m->b_px);
    launch_fill_zero(m->tok1, (int64_t)round_up((int64_t)b * ntok * d, 4) * 4, s);                           // tok1 is free: d(transformer output), zero but for the masked rows
    launch_scatter_by_index(m->g_rows_m, m->idx, b, nm, d, m->tok1, (int64_t)ntok * d, 1, s);
    if ((rc = engine_transformer_backward(e, m->tok1, m->enc1, err)) != VITX_OK) return rc;                   // enc1 = d(tokens) [b, np + 1, d]
    if (m->mpp_edrop > 0.f) launch_dropout(m->enc1, 0, (int64_t)b * ntok * d, m->mpp_edrop, m->mpp_seed, 0u, s);
    if ((rc = engine_embed_backward(e, m->enc1, nullptr, err)) != VITX_OK) return rc;                         // patch_embedding, cls_token, pos_embedding
    return VITX_OK;                                                                                          // (mask_token: no path to the loss, see the header)
  }
  if (m->mae) {
    glue_bwd(m, m->px, m->x_px_T, m->dy_px_T, m->rows_m, P + m->w_px, m->dpred, b * nm, dd, pd, m->g_rows_m, G + m->w_px, G + m->b_px);
    launch_fill_zero(m->g_a, (int64_t)round_up((int64_t)b * np * dd, 4) * 4, s);
    HIPCHK(hipMemcpy2DAsync(m->g_a, (size_t)np * dd * 4, m->g_rows_m, (size_t)nm * dd * 4, (size_t)nm * dd * 4, b, hipMemcpyDeviceToDevice, s));
    if ((rc = engine_transformer_backward(m->dec, m->g_a, m->g_b, err)) != VITX_OK) return rc;               // g_b = d(dec_in) [b, np, dd]
    launch_select_rowsum(m->g_b, nullptr, 0, nm, b, np, dd, m->ws, G + m->mask_tok, s);                      // mask_token: every masked slot
    launch_table_grad(m->g_b, np, m->inv, 0, np, b, np, dd, 0, G + m->dpos, s);                             // decoder_pos_emb rows
    HIPCHK(hipMemcpy2DAsync(m->g_c, (size_t)nu * dd * 4, m->g_b + (int64_t)nm * dd, (size_t)np * dd * 4, (size_t)nu * dd * 4, b, hipMemcpyDeviceToDevice, s));
    const float* d_enc = m->g_c;                                                                            // d(enc_to_dec out) [b, nu, dd]
    if (m->project) {
      glue_bwd(m, m->ed, m->x_ed_T, m->dy_ed_T, m->enc_out, P + m->w_ed, m->g_c, b * nu, d, dd, m->g_a, G + m->w_ed, G + m->b_ed);
      d_enc = m->g_a;
    }
    if ((rc = engine_transformer_backward(e, d_enc, m->g_b, err)) != VITX_OK) return rc;                      // g_b = d(sel) [b, nu, d]
    launch_scatter_rows(m->g_b, nu, m->inv, nm, np, b, np, d, m->g_c, s);                                   // d(tokens) [b, np, d]
    if ((rc = engine_patch_tokens_backward(e, m->g_c, err)) != VITX_OK) return rc;
  } else {
    glue_bwd(m, m->px, m->x_px_T, m->dy_px_T, m->rows_m, P + m->w_px, m->dpred, b * nm, d, pd, m->g_rows_m, G + m->w_px, G + m->b_px);
    launch_scatter_rows(m->g_rows_m, nm, m->inv, 0, nm, b, np, d, m->g_a, s);                               // d(encoded) [b, np, d]
    if ((rc = engine_transformer_backward(e, m->g_a, m->g_b, err)) != VITX_OK) return rc;                    // g_b = d(tokens)
    launch_select_rowsum(m->g_b, m->inv, 0, 0, b, np, d, m->ws, G + m->mask_tok, s);
    launch_gather_rows(m->g_b, (int64_t)np * d, m->idx, nm, 0, b, nm, d, m->g_c, s);                        // masked rows of d(tokens)
    launch_zero_selected_rows(m->g_b, m->inv, (int64_t)b * np, d, s);                                       // tf.where: masked slots ignore the embedding
    if ((rc = engine_patch_tokens_backward(e, m->g_b, err)) != VITX_OK) return rc;
    launch_table_grad(m->g_c, nm, m->inv, 0, nm, b, np, d, 1, e->grads + e->pos + d, s);                    // pos_emb also feeds the mask tokens (simmim.py:103)
  }
  return VITX_OK;
}

int check_indices(const vitx_mim* m, const int32_t* idx, int b, int np, int nm, std::string& err) {
  const int k = m->mae ? np : nm;
  std::vector<char> seen((size_t)np);
  for (int bi = 0; bi < b; ++bi) {
    std::fill(seen.begin(), seen.end(), 0);
    for (int j = 0; j < k; ++j) {
      const int t = idx[(int64_t)bi * k + j];
      if (t < 0 || t >= np) { err = "The values of index must be between 0 and (self.shape[dim] -1)"; return VITX_ERR_INVALID; }   // simmim.py:31
      if (seen[(size_t)t]) { err = "patch indices must be distinct within an image"; return VITX_ERR_INVALID; }
      seen[(size_t)t] = 1;
    }
  }
  return VITX_OK;
}

}  // namespace

#define MIM_TRY try {
#define MIM_CATCH                                                                    \
  }                                                                                  \
  catch (const std::exception& ex) { return capi_fail(VITX_ERR_INVALID, ex.what()); } \
  catch (...) { return capi_fail(VITX_ERR_INVALID, "unknown C++ exception"); }
#define MIM_HIP(x)                                                                                         \
  do {                                                                                                     \
    hipError_t e_ = (x);                                                                                   \
    if (e_ != hipSuccess) return capi_fail(VITX_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));  \
  } while (0)

extern "C" {

int32_t vitx_mim_create(vitx_handle encoder, const vitx_mim_config* cfg, vitx_mim_handle* out) {
  MIM_TRY
  if (!encoder || !cfg || !out) return capi_fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  vitx_mim* m = nullptr;
  int rc = mim_create(encoder, *cfg, &m, err);
  if (rc != VITX_OK) return capi_fail(rc, err);
  *out = m;
  return VITX_OK;
  MIM_CATCH
}
int32_t vitx_mim_destroy(vitx_mim_handle m) {
  MIM_TRY
  mim_destroy(m);
  return VITX_OK;
  MIM_CATCH
}
vitx_handle vitx_mim_decoder(vitx_mim_handle m) { return m ? m->dec : nullptr; }

int32_t vitx_mim_param_table_size(vitx_mim_handle m, int64_t* n_tensors, int64_t* n_elems) {
  if (!m) return capi_fail(VITX_ERR_INVALID, "null handle");
  if (n_tensors) *n_tensors = (int64_t)m->table.size();
  if (n_elems) *n_elems = m->n_params;
  return VITX_OK;
}
int32_t vitx_mim_param_table_entry(vitx_mim_handle m, int64_t index, char* name, int32_t name_cap, int64_t shape[4], int32_t* rank,
                                   int64_t* offset_elems) {
  if (!m) return capi_fail(VITX_ERR_INVALID, "null handle");
  if (index < 0 || index >= (int64_t)m->table.size()) return capi_fail(VITX_ERR_INVALID, "parameter index out of range");
  const ParamDesc& p = m->table[(size_t)index];
  if (name && name_cap > 0) { std::strncpy(name, p.name.c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0; }
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = i < (int)p.shape.size() ? p.shape[(size_t)i] : 1;
  if (rank) *rank = (int32_t)p.shape.size();
  if (offset_elems) *offset_elems = p.offset;
  return VITX_OK;
}

static int mim_copy_blob(vitx_mim* m, float* arena, float* host, int64_t n, bool to_device) {
  if (n != m->n_params) return capi_fail(VITX_ERR_INVALID, "blob size does not match the wrapper's parameter table");
  hipStream_t s = m->enc->stream;
  for (auto& p : m->table) {
    if (to_device) MIM_HIP(hipMemcpyAsync(arena + p.aoff, host + p.offset, (size_t)p.count * 4, hipMemcpyHostToDevice, s));
    else MIM_HIP(hipMemcpyAsync(host + p.offset, arena + p.aoff, (size_t)p.count * 4, hipMemcpyDeviceToHost, s));
  }
  MIM_HIP(hipStreamSynchronize(s));
  return VITX_OK;
}
int32_t vitx_mim_set_params(vitx_mim_handle m, const float* host_blob, int64_t n) {
  MIM_TRY
  if (!m || !host_blob) return capi_fail(VITX_ERR_INVALID, "null argument");
  m->w_dirty = true;
  return mim_copy_blob(m, m->params, const_cast<float*>(host_blob), n, true);
  MIM_CATCH
}
int32_t vitx_mim_get_params(vitx_mim_handle m, float* host_blob, int64_t n) {
  MIM_TRY
  if (!m || !host_blob) return capi_fail(VITX_ERR_INVALID, "null argument");
  return mim_copy_blob(m, m->params, host_blob, n, false);
  MIM_CATCH
}
int32_t vitx_mim_get_grads(vitx_mim_handle m, float* host_blob, int64_t n) {
  MIM_TRY
  if (!m || !host_blob) return capi_fail(VITX_ERR_INVALID, "null argument");
  return mim_copy_blob(m, m->grads, host_blob, n, false);
  MIM_CATCH
}
int32_t vitx_mim_params_dev(vitx_mim_handle m, float** params_dev, float** grads_dev, int64_t* n_elems) {
  if (!m) return capi_fail(VITX_ERR_INVALID, "null handle");
  if (params_dev) *params_dev = m->params;
  if (grads_dev) *grads_dev = m->grads;
  if (n_elems) *n_elems = m->n_arena;
  return VITX_OK;
}

int32_t vitx_mim_params_changed(vitx_mim_handle m) {
  if (!m) return capi_fail(VITX_ERR_INVALID, "null handle");
  m->w_dirty = true;
  return VITX_OK;
}

int32_t vitx_mim_num_masked(vitx_mim_handle m, int32_t H, int32_t W, int32_t* num_patches, int32_t* num_masked) {
  if (!m) return capi_fail(VITX_ERR_INVALID, "null handle");
  const vitx_config& c = m->enc->cfg;
  if (H <= 0 || W <= 0 || H % c.patch_h || W % c.patch_w) return capi_fail(VITX_ERR_INVALID, "Image dimensions must be divisible by the patch size.");
  const int np = (H / c.patch_h) * (W / c.patch_w);
  if (num_patches) *num_patches = np;
  if (num_masked) *num_masked = mim_num_masked(m, np);
  return VITX_OK;
}

int32_t vitx_mim_forward_dev(vitx_mim_handle m, const float* img_dev, int32_t b, int32_t H, int32_t W, const int32_t* idx_dev, int32_t training, uint64_t seed, float* loss_dev) {
  MIM_TRY
  if (!m || !img_dev || !idx_dev) return capi_fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  int rc = mim_forward(m, img_dev, b, H, W, idx_dev, training, seed, err);
  if (rc != VITX_OK) return capi_fail(rc, err);
  if (loss_dev) MIM_HIP(hipMemcpyAsync(loss_dev, m->loss, 4, hipMemcpyDeviceToDevice, m->enc->stream));
  return VITX_OK;
  MIM_CATCH
}
int32_t vitx_mim_forward(vitx_mim_handle m, const float* img_host, int32_t b, int32_t H, int32_t W, const int32_t* idx_host, int32_t training, uint64_t seed, float* loss_host) {
  MIM_TRY
  if (!m || !img_host || !idx_host) return capi_fail(VITX_ERR_INVALID, "null argument");
  const vitx_config& c = m->enc->cfg;
  if (b <= 0 || b > c.max_batch) return capi_fail(VITX_ERR_INVALID, "batch must be in [1, max_batch]");
  if (H <= 0 || W <= 0 || H > c.image_h || W > c.image_w || H % c.patch_h || W % c.patch_w)
    return capi_fail(VITX_ERR_INVALID, "Image dimensions must be divisible by the patch size.");
  const int np = (H / c.patch_h) * (W / c.patch_w), nm = mim_num_masked(m, np);
  std::string err;
  int rc = check_indices(m, idx_host, b, np, nm, err);
  if (rc != VITX_OK) return capi_fail(rc, err);
  hipStream_t s = m->enc->stream;
  MIM_HIP(hipMemcpyAsync(m->img, img_host, (size_t)b * H * W * c.channels * 4, hipMemcpyHostToDevice, s));
  MIM_HIP(hipMemcpyAsync(m->idx, idx_host, (size_t)b * (m->mae ? np : nm) * 4, hipMemcpyHostToDevice, s));
  rc = mim_forward(m, m->img, b, H, W, m->idx, training, seed, err);
  if (rc != VITX_OK) return capi_fail(rc, err);
  if (loss_host) MIM_HIP(hipMemcpyAsync(loss_host, m->loss, 4, hipMemcpyDeviceToHost, s));
  MIM_HIP(hipStreamSynchronize(s));
  return VITX_OK;
  MIM_CATCH
}
int32_t vitx_mim_backward(vitx_mim_handle m) {
  MIM_TRY
  if (!m) return capi_fail(VITX_ERR_INVALID, "null handle");
  std::string err;
  int rc = mim_backward(m, err);
  if (rc != VITX_OK) return capi_fail(rc, err);
  return VITX_OK;
  MIM_CATCH
}

int32_t vitx_mim_read(vitx_mim_handle m, const char* which, float* out_host, int64_t cap, int64_t* n_elems) {
  MIM_TRY
  if (!m || !which || !out_host) return capi_fail(VITX_ERR_INVALID, "null argument");
  if (!m->have_fwd) return capi_fail(VITX_ERR_STATE, "read requires a preceding forward");
  const std::string w = which;
  const float* src = nullptr;
  int64_t n = 0;
  const int64_t b = m->b, np = m->np, nm = m->nm;
  if (w == "pred") { src = m->pred; n = b * nm * m->po; }
  else if (w == "target") { src = m->target; n = b * nm * m->pd; }
  else if (w == "patches") { src = m->patches; n = b * np * m->pd; }
  else if (w == "encoded" && m->mpp) { src = m->enc1; n = b * (np + 1) * m->d; }
  else if (w == "encoded") { src = m->enc_out; n = b * (m->mae ? np - nm : np) * m->d; }
  else if (w == "decoded" && m->mae) { src = m->dec_out; n = b * np * m->dd; }
  else return capi_fail(VITX_ERR_INVALID, "unknown tensor name");
  if (n_elems) *n_elems = n;
  if (n > cap) return capi_fail(VITX_ERR_INVALID, "output buffer too small");
  hipStream_t s = m->enc->stream;
  MIM_HIP(hipMemcpyAsync(out_host, src, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  MIM_HIP(hipStreamSynchronize(s));
  return VITX_OK;
  MIM_CATCH
}

}  // extern "C"
