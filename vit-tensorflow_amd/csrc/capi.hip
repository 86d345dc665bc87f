// extern "C" surface declared in include/vitx.h.  No exceptions cross this boundary.
#include <dlfcn.h>

#include <cstring>
#include <map>

#include "engine.h"

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
int capi_fail(int code, const std::string& msg) { return fail(code, msg); }   // shared with mim.hip
#define CAPI_HIP(x)                                                                                    \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) return fail(VITX_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));   \
  } while (0)
#define CAPI_TRY try {
#define CAPI_CATCH                                                       \
  }                                                                      \
  catch (const std::exception& ex) { return fail(VITX_ERR_INVALID, ex.what()); } \
  catch (...) { return fail(VITX_ERR_INVALID, "unknown C++ exception"); }

extern "C" {

const char* vitx_version(void) { return "vitx 0.1.0 (gfx950)"; }
const char* vitx_last_error(void) { return g_last_error.c_str(); }

int32_t vitx_param_table_size(const vitx_config* cfg, int64_t* n_tensors, int64_t* n_elems) {
  CAPI_TRY
  if (!cfg) return fail(VITX_ERR_INVALID, "null config");
  vitx_config c = *cfg;
  if (c.channels <= 0) c.channels = 3;
  std::vector<ParamDesc> t;
  std::string err = build_param_table(c, t);
  if (!err.empty()) return fail(VITX_ERR_INVALID, err);
  if (n_tensors) *n_tensors = (int64_t)t.size();
  if (n_elems) *n_elems = t.back().offset + t.back().count;
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_param_table_entry(const vitx_config* cfg, int64_t index, char* name, int32_t name_cap, int64_t shape[4], int32_t* rank,
                               int64_t* offset_elems) {
  CAPI_TRY
  if (!cfg) return fail(VITX_ERR_INVALID, "null config");
  vitx_config c = *cfg;
  if (c.channels <= 0) c.channels = 3;
  std::vector<ParamDesc> t;
  std::string err = build_param_table(c, t);
  if (!err.empty()) return fail(VITX_ERR_INVALID, err);
  if (index < 0 || index >= (int64_t)t.size()) return fail(VITX_ERR_INVALID, "parameter index out of range");
  const ParamDesc& p = t[(size_t)index];
  if (name && name_cap > 0) {
    std::strncpy(name, p.name.c_str(), (size_t)name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = i < (int)p.shape.size() ? p.shape[(size_t)i] : 1;
  if (rank) *rank = (int32_t)p.shape.size();
  if (offset_elems) *offset_elems = p.offset;
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_create(const vitx_config* cfg, vitx_handle* out) {
  CAPI_TRY
  if (!cfg || !out) return fail(VITX_ERR_INVALID, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(VITX_ERR_HIP, "no HIP device available: libvitx has no CPU fallback (the product path is HIP-only)");
  std::string err;
  vitx_engine* e = nullptr;
  int rc = engine_create(*cfg, &e, err);
  if (rc != VITX_OK) return fail(rc, err);
  *out = e;
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_get_config(vitx_handle h, vitx_config* out) {
  if (!h || !out) return fail(VITX_ERR_INVALID, "null argument");
  *out = h->cfg;
  return VITX_OK;
}

int32_t vitx_destroy(vitx_handle h) {
  CAPI_TRY
  engine_destroy(h);
  return VITX_OK;
  CAPI_CATCH
}

// packed host blob <-> aligned device arena
static int copy_blob(vitx_engine* e, float* arena, float* host, int64_t n, bool to_device) {
  if (n != e->n_params) return fail(VITX_ERR_INVALID, "blob size does not match the parameter table");
  if (e->n_arena == e->n_params) {
    if (to_device) CAPI_HIP(hipMemcpyAsync(arena, host, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    else CAPI_HIP(hipMemcpyAsync(host, arena, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream));
  } else {
    for (auto& p : e->table) {
      if (to_device) CAPI_HIP(hipMemcpyAsync(arena + p.aoff, host + p.offset, (size_t)p.count * 4, hipMemcpyHostToDevice, e->stream));
      else CAPI_HIP(hipMemcpyAsync(host + p.offset, arena + p.aoff, (size_t)p.count * 4, hipMemcpyDeviceToHost, e->stream));
    }
  }
  CAPI_HIP(hipStreamSynchronize(e->stream));
  return VITX_OK;
}

int32_t vitx_set_params(vitx_handle h, const float* host_blob, int64_t n) {
  CAPI_TRY
  if (!h || !host_blob) return fail(VITX_ERR_INVALID, "null argument");
  int rc = copy_blob(h, h->params, const_cast<float*>(host_blob), n, true);
  h->params_dirty = true;
  return rc;
  CAPI_CATCH
}
int32_t vitx_get_params(vitx_handle h, float* host_blob, int64_t n) {
  CAPI_TRY
  if (!h || !host_blob) return fail(VITX_ERR_INVALID, "null argument");
  return copy_blob(h, h->params, host_blob, n, false);
  CAPI_CATCH
}
int32_t vitx_get_grads(vitx_handle h, float* host_blob, int64_t n) {
  CAPI_TRY
  if (!h || !host_blob) return fail(VITX_ERR_INVALID, "null argument");
  return copy_blob(h, h->grads, host_blob, n, false);
  CAPI_CATCH
}
int32_t vitx_params_dev(vitx_handle h, float** p, int64_t* n) {
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  if (p) *p = h->params;
  if (n) *n = h->n_arena;
  return VITX_OK;
}
int32_t vitx_grads_dev(vitx_handle h, float** p, int64_t* n) {
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  if (p) *p = h->grads;
  if (n) *n = h->n_arena;
  return VITX_OK;
}
int32_t vitx_bind_arenas(vitx_handle h, float* params_dev, float* grads_dev) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  if (params_dev && params_dev != h->params) {
    CAPI_HIP(hipMemcpyAsync(params_dev, h->params, (size_t)h->n_arena * 4, hipMemcpyDeviceToDevice, h->stream));
    CAPI_HIP(hipStreamSynchronize(h->stream));
    h->params = params_dev;
    engine_params_moved(h);   // (round 6) the batched operand refresh reads the arena through a device table of absolute pointers
  }
  if (grads_dev && grads_dev != h->grads) {
    CAPI_HIP(hipMemsetAsync(grads_dev, 0, (size_t)h->n_arena * 4, h->stream));
    CAPI_HIP(hipStreamSynchronize(h->stream));
    h->grads = grads_dev;
  }
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_params_changed(vitx_handle h) {
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  h->params_dirty = true;
  return VITX_OK;
}

int32_t vitx_forward_dev(vitx_handle h, const float* img_dev, int32_t b, int32_t H, int32_t W, int32_t training, uint64_t seed,
                         float* logits_dev) {
  CAPI_TRY
  if (!h || !img_dev) return fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  int rc = engine_forward(h, img_dev, b, H, W, training, seed, logits_dev, err);
  if (rc != VITX_OK) return fail(rc, err);
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_set_patch_input(vitx_handle h, int32_t np) {
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  if (np < 0 || np > h->np_max) return fail(VITX_ERR_INVALID, "set_patch_input: np must be in [0, num_patches]");
  h->next_patch_np = np;
  return VITX_OK;
}

int32_t vitx_forward(vitx_handle h, const float* img_host, int32_t b, int32_t H, int32_t W, int32_t training, uint64_t seed,
                     float* logits_host) {
  CAPI_TRY
  // the one-shot vitx_set_patch_input state is consumed FIRST: a call that fails validation must not leave the handle armed (the next ordinary
  // forward would read its image as patch rows)
  const int np_once = h ? h->next_patch_np : 0;
  if (h) h->next_patch_np = 0;
  if (!h || !img_host || !logits_host) return fail(VITX_ERR_INVALID, "null argument");
  if (np_once > 0) return vitx_forward_patches(h, img_host, b, np_once, training, seed, logits_host);   // img_host holds patch rows [b, np, patch_dim]
  if (b <= 0 || b > h->cfg.max_batch) return fail(VITX_ERR_INVALID, "batch must be in [1, max_batch]");
  if (H <= 0 || W <= 0 || H > h->cfg.image_h || W > h->cfg.image_w)
    return fail(VITX_ERR_INVALID, "image larger than the configured image_size");
  CAPI_HIP(hipMemcpyAsync(h->img_dev, img_host, (size_t)b * H * W * h->cfg.channels * 4, hipMemcpyHostToDevice, h->stream));
  std::string err;
  int rc = engine_forward(h, h->img_dev, b, H, W, training, seed, nullptr, err);
  if (rc != VITX_OK) return fail(rc, err);
  CAPI_HIP(hipMemcpy2DAsync(logits_host, (size_t)h->cfg.num_classes * 4, h->logits, (size_t)h->nc_k * 4, (size_t)h->cfg.num_classes * 4,
                            (size_t)b, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}

// Forward from caller-supplied patch rows [b, np, patch_dim] (fp32) instead of an image: the patch Dense, cls / position rows, the
// transformer and the head run as usual.  vitx_backward(_dev) after it returns d(patches) [b, np, patch_dim] where it would return
// d(img).  T2T-ViT's patch_embedding is a tokenizer pipeline that ends in this Dense (t2t.py:59-77).
int32_t vitx_forward_patches_dev(vitx_handle h, const float* patches_dev, int32_t b, int32_t np, int32_t training, uint64_t seed,
                                 float* logits_dev) {
  CAPI_TRY
  if (!h || !patches_dev) return fail(VITX_ERR_INVALID, "null argument");
  h->fwd_patches = patches_dev;
  h->fwd_np = np;
  std::string err;
  int rc = engine_forward(h, nullptr, b, 0, 0, training, seed, logits_dev, err);
  h->fwd_patches = nullptr;
  if (rc != VITX_OK) return fail(rc, err);
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_forward_patches(vitx_handle h, const float* patches_host, int32_t b, int32_t np, int32_t training, uint64_t seed,
                             float* logits_host) {
  CAPI_TRY
  if (!h || !patches_host || !logits_host) return fail(VITX_ERR_INVALID, "null argument");
  if (b <= 0 || b > h->cfg.max_batch) return fail(VITX_ERR_INVALID, "batch must be in [1, max_batch]");
  if (np <= 0 || np > h->np_max) return fail(VITX_ERR_INVALID, "forward_patches: np must be in [1, num_patches]");
  CAPI_HIP(hipMemcpyAsync(h->img_dev, patches_host, (size_t)b * np * h->pd * 4, hipMemcpyHostToDevice, h->stream));
  h->fwd_patches = h->img_dev;
  h->fwd_np = np;
  std::string err;
  int rc = engine_forward(h, nullptr, b, 0, 0, training, seed, nullptr, err);
  h->fwd_patches = nullptr;
  if (rc != VITX_OK) return fail(rc, err);
  CAPI_HIP(hipMemcpy2DAsync(logits_host, (size_t)h->cfg.num_classes * 4, h->logits, (size_t)h->nc_k * 4, (size_t)h->cfg.num_classes * 4,
                            (size_t)b, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_backward_dev(vitx_handle h, const float* dlogits_dev, float* dimg_dev) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  std::string err;
  int rc = engine_backward(h, dlogits_dev, dimg_dev, err);
  if (rc != VITX_OK) return fail(rc, err);
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_backward(vitx_handle h, const float* dlogits_host, float* dimg_host) {
  CAPI_TRY
  if (!h || !dlogits_host) return fail(VITX_ERR_INVALID, "null argument");
  if (!h->have_fwd) return fail(VITX_ERR_STATE, "backward requires a preceding forward");
  const int b = h->last_b, nc = h->cfg.num_classes;
  CAPI_HIP(hipMemcpy2DAsync(h->dlogits, (size_t)h->nc_k * 4, dlogits_host, (size_t)nc * 4, (size_t)nc * 4, (size_t)b, hipMemcpyHostToDevice,
                            h->stream));
  float* dimg_dev = nullptr;
  const size_t img_bytes = (size_t)b * h->last_H * h->last_W * h->cfg.channels * 4;
  if (dimg_host) dimg_dev = h->img_dev;   // the staged image is no longer needed once patches are unfolded
  std::string err;
  int rc = engine_backward(h, nullptr, dimg_dev, err);
  if (rc != VITX_OK) return fail(rc, err);
  if (dimg_host) CAPI_HIP(hipMemcpyAsync(dimg_host, dimg_dev, img_bytes, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_transformer_forward(vitx_handle h, const float* tokens_host, int32_t b, int32_t n, int32_t training, uint64_t seed, float* out_host) {
  CAPI_TRY
  if (!h || !tokens_host || !out_host) return fail(VITX_ERR_INVALID, "null argument");
  if (b <= 0 || b > h->cfg.max_batch || n <= 0 || n > h->ntok_cap) return fail(VITX_ERR_INVALID, "transformer_forward: b or n out of range");
  const size_t bytes = (size_t)b * n * h->cfg.dim * 4;
  float* tmp = h->g;   // [>= mp, d] fp32 scratch that no forward kernel touches
  CAPI_HIP(hipMemcpyAsync(tmp, tokens_host, bytes, hipMemcpyHostToDevice, h->stream));
  std::string err;
  int rc = engine_transformer_forward(h, tmp, b, n, training, seed, tmp, err);
  if (rc != VITX_OK) return fail(rc, err);
  CAPI_HIP(hipMemcpyAsync(out_host, tmp, bytes, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_transformer_backward(vitx_handle h, const float* dout_host, float* dtokens_host_or_null) {
  CAPI_TRY
  if (!h || !dout_host) return fail(VITX_ERR_INVALID, "null argument");
  if (!h->have_tf) return fail(VITX_ERR_STATE, "transformer_backward requires a preceding transformer_forward");
  const size_t bytes = (size_t)h->tf_b * h->tf_n * h->cfg.dim * 4;
  float* tmp = h->tmp_f32;   // [>= mp, max(d, pd)] fp32 scratch; block_backward overwrites it only after d(out) has been consumed
  CAPI_HIP(hipMemcpyAsync(tmp, dout_host, bytes, hipMemcpyHostToDevice, h->stream));
  std::string err;
  int rc = engine_transformer_backward(h, tmp, dtokens_host_or_null ? tmp : nullptr, err);
  if (rc != VITX_OK) return fail(rc, err);
  if (dtokens_host_or_null) CAPI_HIP(hipMemcpyAsync(dtokens_host_or_null, tmp, bytes, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_patch_unfold(const float* img_host, int32_t b, int32_t H, int32_t W, int32_t C, int32_t ph, int32_t pw, float* out_host) {
  CAPI_TRY
  if (!img_host || !out_host) return fail(VITX_ERR_INVALID, "null argument");
  if (b < 0 || H <= 0 || W <= 0 || C <= 0 || ph <= 0 || pw <= 0) return fail(VITX_ERR_INVALID, "sizes must be positive");
  if (H % ph || W % pw) return fail(VITX_ERR_INVALID, "Image dimensions must be divisible by the patch size.");
  if (b == 0) return VITX_OK;   // empty batch: nothing to do
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(VITX_ERR_HIP, "no HIP device available (no CPU fallback)");
  const size_t n = (size_t)b * H * W * C;
  float *din = nullptr, *dout = nullptr;
  struct Free { float*& p; ~Free() { if (p) (void)hipFree(p); } } free_in{din}, free_out{dout};   // every exit path releases both
  CAPI_HIP(hipMalloc((void**)&din, n * 4));
  CAPI_HIP(hipMalloc((void**)&dout, n * 4));
  CAPI_HIP(hipMemcpy(din, img_host, n * 4, hipMemcpyHostToDevice));
  launch_unfold(din, dout, 0, b, H, W, C, ph, pw, (int64_t)ph * pw * C, nullptr);
  CAPI_HIP(hipMemcpy(out_host, dout, n * 4, hipMemcpyDeviceToHost));
  return VITX_OK;
  CAPI_CATCH
}

// ---- efficient.ViT shell (efficient.py:12-56)
int32_t vitx_embed_forward_dev(vitx_handle h, const float* img_dev, int32_t b, int32_t H, int32_t W, float* tokens_dev) {
  CAPI_TRY
  if (!h || !img_dev || !tokens_dev) return fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  int rc = engine_embed_forward(h, img_dev, b, H, W, tokens_dev, err);
  if (rc != VITX_OK) return fail(rc, err);
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_embed_forward(vitx_handle h, const float* img_host, int32_t b, int32_t H, int32_t W, float* tokens_host) {
  CAPI_TRY
  if (!h || !img_host || !tokens_host) return fail(VITX_ERR_INVALID, "null argument");
  if (b <= 0 || b > h->cfg.max_batch) return fail(VITX_ERR_INVALID, "batch must be in [1, max_batch]");
  if (H <= 0 || W <= 0 || H > h->cfg.image_h || W > h->cfg.image_w) return fail(VITX_ERR_INVALID, "image larger than the configured image_size");
  CAPI_HIP(hipMemcpyAsync(h->img_dev, img_host, (size_t)b * H * W * h->cfg.channels * 4, hipMemcpyHostToDevice, h->stream));
  std::string err;
  float* tmp = h->tmp_f32;   // [>= mp, max(d, pd)] fp32 scratch
  int rc = engine_embed_forward(h, h->img_dev, b, H, W, tmp, err);
  if (rc != VITX_OK) return fail(rc, err);
  CAPI_HIP(hipMemcpyAsync(tokens_host, tmp, (size_t)b * h->last_ntok * h->cfg.dim * 4, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_patch_dense_forward(vitx_handle h, const float* patches_host, int32_t rows, float* out_host) {
  CAPI_TRY
  if (!h || !patches_host || !out_host) return fail(VITX_ERR_INVALID, "null argument");
  if (rows <= 0 || (int64_t)rows > (int64_t)h->cfg.max_batch * h->np_max)
    return fail(VITX_ERR_INVALID, "patch_dense_forward: rows must be in [1, max_batch * num_patches]");
  CAPI_HIP(hipMemcpyAsync(h->tmp_f32, patches_host, (size_t)rows * h->pd * 4, hipMemcpyHostToDevice, h->stream));
  std::string err;
  int rc = engine_patch_dense_forward(h, h->tmp_f32, rows, h->g, err);
  if (rc != VITX_OK) return fail(rc, err);
  CAPI_HIP(hipMemcpyAsync(out_host, h->g, (size_t)rows * h->cfg.dim * 4, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_head_forward_dev(vitx_handle h, const float* x_dev, int32_t b, int32_t n, float* logits_dev) {
  CAPI_TRY
  if (!h || !x_dev) return fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  int rc = engine_head_forward(h, x_dev, b, n, logits_dev, err);
  if (rc != VITX_OK) return fail(rc, err);
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_head_forward(vitx_handle h, const float* x_host, int32_t b, int32_t n, float* logits_host) {
  CAPI_TRY
  if (!h || !x_host || !logits_host) return fail(VITX_ERR_INVALID, "null argument");
  if (b <= 0 || b > h->cfg.max_batch || n <= 0 || n > h->ntok_cap) return fail(VITX_ERR_INVALID, "head_forward: b or n out of range");
  float* tmp = h->tmp_f32;
  CAPI_HIP(hipMemcpyAsync(tmp, x_host, (size_t)b * n * h->cfg.dim * 4, hipMemcpyHostToDevice, h->stream));
  std::string err;
  int rc = engine_head_forward(h, tmp, b, n, nullptr, err);
  if (rc != VITX_OK) return fail(rc, err);
  CAPI_HIP(hipMemcpy2DAsync(logits_host, (size_t)h->cfg.num_classes * 4, h->logits, (size_t)h->nc_k * 4, (size_t)h->cfg.num_classes * 4,
                            (size_t)b, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_head_backward_dev(vitx_handle h, const float* dlogits_dev_or_null, float* dx_dev) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  std::string err;
  int rc = engine_head_backward(h, dlogits_dev_or_null, dx_dev, err);
  if (rc != VITX_OK) return fail(rc, err);
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_head_backward(vitx_handle h, const float* dlogits_host, float* dx_host) {
  CAPI_TRY
  if (!h || !dlogits_host || !dx_host) return fail(VITX_ERR_INVALID, "null argument");
  if (!h->have_head) return fail(VITX_ERR_STATE, "head_backward requires a preceding head_forward");
  const int b = h->shell_b, nc = h->cfg.num_classes;
  CAPI_HIP(hipMemcpy2DAsync(h->dlogits, (size_t)h->nc_k * 4, dlogits_host, (size_t)nc * 4, (size_t)nc * 4, (size_t)b, hipMemcpyHostToDevice,
                            h->stream));
  std::string err;
  int rc = engine_head_backward(h, nullptr, nullptr, err);
  if (rc != VITX_OK) return fail(rc, err);
  CAPI_HIP(hipMemcpyAsync(dx_host, h->g, (size_t)b * h->shell_n * h->cfg.dim * 4, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_embed_backward_dev(vitx_handle h, const float* dtokens_dev, float* dimg_dev_or_null) {
  CAPI_TRY
  if (!h || !dtokens_dev) return fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  int rc = engine_embed_backward(h, dtokens_dev, dimg_dev_or_null, err);
  if (rc != VITX_OK) return fail(rc, err);
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_embed_backward(vitx_handle h, const float* dtokens_host, float* dimg_host_or_null) {
  CAPI_TRY
  if (!h || !dtokens_host) return fail(VITX_ERR_INVALID, "null argument");
  if (!h->have_embed) return fail(VITX_ERR_STATE, "embed_backward requires a preceding embed_forward");
  const int b = h->last_b;
  CAPI_HIP(hipMemcpyAsync(h->g, dtokens_host, (size_t)b * h->last_ntok * h->cfg.dim * 4, hipMemcpyHostToDevice, h->stream));
  float* dimg_dev = dimg_host_or_null ? h->img_dev : nullptr;   // the staged image is no longer needed once patches are unfolded
  std::string err;
  int rc = engine_embed_backward(h, h->g, dimg_dev, err);
  if (rc != VITX_OK) return fail(rc, err);
  if (dimg_host_or_null)
    CAPI_HIP(hipMemcpyAsync(dimg_host_or_null, dimg_dev, (size_t)b * h->last_H * h->last_W * h->cfg.channels * 4, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
  CAPI_CATCH
}

// ---- T2T tokenizer: tf.image.extract_patches(..., padding='SAME') (t2t.py:42)
static int extract_patches_check(int b, int H, int W, int C, int k, int st) {
  if (b < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || st <= 0) return fail(VITX_ERR_INVALID, "sizes must be positive");
  return VITX_OK;
}
int32_t vitx_extract_patches_shape(int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t* oh, int32_t* ow, int32_t* feat) {
  CAPI_TRY
  int rc = extract_patches_check(0, H, W, C, k, stride);
  if (rc != VITX_OK) return rc;
  int a, bb, pt, pl;
  extract_patches_geometry(H, W, k, stride, &a, &bb, &pt, &pl);
  if (oh) *oh = a;
  if (ow) *ow = bb;
  if (feat) *feat = k * k * C;
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_extract_patches_dev(const float* x_dev, int32_t b, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, float* out_dev,
                                 void* hip_stream) {
  CAPI_TRY
  if (!x_dev || !out_dev) return fail(VITX_ERR_INVALID, "null argument");
  int rc = extract_patches_check(b, H, W, C, k, stride);
  if (rc != VITX_OK) return rc;
  launch_extract_patches(x_dev, out_dev, b, H, W, C, k, stride, (hipStream_t)hip_stream);
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_extract_patches_backward_dev(const float* dout_dev, int32_t b, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride,
                                          float* dx_dev, void* hip_stream) {
  CAPI_TRY
  if (!dout_dev || !dx_dev) return fail(VITX_ERR_INVALID, "null argument");
  int rc = extract_patches_check(b, H, W, C, k, stride);
  if (rc != VITX_OK) return rc;
  launch_extract_patches_bwd(dout_dev, dx_dev, b, H, W, C, k, stride, (hipStream_t)hip_stream);
  return VITX_OK;
  CAPI_CATCH
}
static int extract_patches_host(const float* in_host, float* out_host, int b, int H, int W, int C, int k, int st, bool backward) {
  int rc = extract_patches_check(b, H, W, C, k, st);
  if (rc != VITX_OK) return rc;
  if (b == 0) return VITX_OK;   // empty batch: nothing to do
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(VITX_ERR_HIP, "no HIP device available (no CPU fallback)");
  int oh, ow, pt, pl;
  extract_patches_geometry(H, W, k, st, &oh, &ow, &pt, &pl);
  const size_t nx = (size_t)b * H * W * C, no = (size_t)b * oh * ow * k * k * C;
  const size_t nin = backward ? no : nx, nout = backward ? nx : no;
  float *din = nullptr, *dout = nullptr;
  CAPI_HIP(hipMalloc((void**)&din, nin * 4));
  if (hipMalloc((void**)&dout, nout * 4) != hipSuccess) { (void)hipFree(din); return fail(VITX_ERR_HIP, "hipMalloc failed"); }
  hipError_t e1 = hipMemcpy(din, in_host, nin * 4, hipMemcpyHostToDevice);
  if (e1 == hipSuccess) {
    if (backward) launch_extract_patches_bwd(din, dout, b, H, W, C, k, st, nullptr);
    else launch_extract_patches(din, dout, b, H, W, C, k, st, nullptr);
    e1 = hipMemcpy(out_host, dout, nout * 4, hipMemcpyDeviceToHost);
  }
  (void)hipFree(din);
  (void)hipFree(dout);
  if (e1 != hipSuccess) return fail(VITX_ERR_HIP, std::string("extract_patches: ") + hipGetErrorString(e1));
  return VITX_OK;
}
int32_t vitx_extract_patches(const float* x_host, int32_t b, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, float* out_host) {
  CAPI_TRY
  if (!x_host || !out_host) return fail(VITX_ERR_INVALID, "null argument");
  return extract_patches_host(x_host, out_host, b, H, W, C, k, stride, false);
  CAPI_CATCH
}
int32_t vitx_extract_patches_backward(const float* dout_host, int32_t b, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride,
                                      float* dx_host) {
  CAPI_TRY
  if (!dout_host || !dx_host) return fail(VITX_ERR_INVALID, "null argument");
  return extract_patches_host(dout_host, dx_host, b, H, W, C, k, stride, true);
  CAPI_CATCH
}

int32_t vitx_ce_loss_grad_dev(vitx_handle h, const int32_t* labels_dev, float inv_global_batch, float* loss_dev) {
  CAPI_TRY
  if (!h || !labels_dev) return fail(VITX_ERR_INVALID, "null argument");
  if (!h->have_fwd) return fail(VITX_ERR_STATE, "loss gradient requires a preceding forward");
  launch_ce_grad(h->logits, h->nc_k, labels_dev, h->last_b, h->cfg.num_classes, inv_global_batch, h->dlogits, h->loss_rows, h->stream);
  if (loss_dev) launch_sum_rows(h->loss_rows, h->last_b, 1, loss_dev, h->stream);
  return VITX_OK;
  CAPI_CATCH
}

static int ensure_opt_state(vitx_engine* e, bool need_v) {
  if (!e->opt_m) {
    CAPI_HIP(hipMalloc((void**)&e->opt_m, (size_t)e->n_arena * 4));
    CAPI_HIP(hipMemsetAsync(e->opt_m, 0, (size_t)e->n_arena * 4, e->stream));
    e->allocs.push_back(e->opt_m);
  }
  if (need_v && !e->opt_v) {
    CAPI_HIP(hipMalloc((void**)&e->opt_v, (size_t)e->n_arena * 4));
    CAPI_HIP(hipMemsetAsync(e->opt_v, 0, (size_t)e->n_arena * 4, e->stream));
    e->allocs.push_back(e->opt_v);
  }
  return VITX_OK;
}

int32_t vitx_adamw_step(vitx_handle h, float lr, float beta1, float beta2, float eps, float weight_decay) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  int rc = ensure_opt_state(h, true);
  if (rc != VITX_OK) return rc;
  h->opt_step += 1;
  launch_adamw(h->params, h->grads, h->opt_m, h->opt_v, h->n_arena, lr, beta1, beta2, eps, weight_decay, h->opt_step, h->stream);
  h->params_dirty = true;
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_sgd_step(vitx_handle h, float lr, float momentum, float weight_decay) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  if (momentum != 0.f) { int rc = ensure_opt_state(h, false); if (rc != VITX_OK) return rc; }
  launch_sgd(h->params, h->grads, momentum != 0.f ? h->opt_m : nullptr, h->n_arena, lr, momentum, weight_decay, h->stream);
  h->params_dirty = true;
  return VITX_OK;
  CAPI_CATCH
}

// Optimizer state (first / second moments in the parameter-arena layout, step count): lives only inside the handle, so whoever
// rebuilds a handle (the Python front grows it when a larger batch arrives) carries it over with this pair.
int32_t vitx_get_opt_state(vitx_handle h, float* m_host, float* v_host, int64_t n_elems, int64_t* step, int32_t* have_m, int32_t* have_v) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  if (step) *step = h->opt_step;
  if (have_m) *have_m = h->opt_m != nullptr;
  if (have_v) *have_v = h->opt_v != nullptr;
  if ((m_host || v_host) && n_elems != h->n_params) return fail(VITX_ERR_INVALID, "optimizer state size mismatch");
  CAPI_HIP(hipStreamSynchronize(h->stream));
  for (int which = 0; which < 2; ++which) {
    float* dst = which ? v_host : m_host;
    const float* src = which ? h->opt_v : h->opt_m;
    if (!dst || !src) continue;
    for (auto& p : h->table) CAPI_HIP(hipMemcpy(dst + p.offset, src + p.aoff, (size_t)p.count * 4, hipMemcpyDeviceToHost));
  }
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_set_opt_state(vitx_handle h, const float* m_host, const float* v_host, int64_t n_elems, int64_t step) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  if ((m_host || v_host) && n_elems != h->n_params) return fail(VITX_ERR_INVALID, "optimizer state size mismatch");
  if (m_host || v_host) { int rc = ensure_opt_state(h, v_host != nullptr); if (rc != VITX_OK) return rc; }
  CAPI_HIP(hipStreamSynchronize(h->stream));
  for (int which = 0; which < 2; ++which) {
    const float* src = which ? v_host : m_host;
    float* dst = which ? h->opt_v : h->opt_m;
    if (!src || !dst) continue;
    for (auto& p : h->table) CAPI_HIP(hipMemcpy(dst + p.aoff, src + p.offset, (size_t)p.count * 4, hipMemcpyHostToDevice));
  }
  h->opt_step = (int)step;
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_set_stream(vitx_handle h, void* s) {
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  (void)hipStreamSynchronize(h->stream);
  h->stream = s ? (hipStream_t)s : h->own_stream;
  return VITX_OK;
}
// ---- HIP graphs: the launch sequence of a step (operand refresh, forward, loss gradient, backward, optimizer) is fixed once the
// batch geometry is, so it can be captured from the handle's stream and replayed with ONE launch: at small batches (the
// reference's README example runs b = 1) the step is bound by the ~300 kernel launches, not by the kernels.
int32_t vitx_graph_capture_begin(vitx_handle h) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  if (h->grad_cb) return fail(VITX_ERR_STATE, "graph capture cannot record the host-side gradient-ready callback: unregister it first");
  if (h->profiling) return fail(VITX_ERR_STATE, "graph capture while profiling");
  CAPI_HIP(hipStreamSynchronize(h->stream));
  CAPI_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_graph_capture_end(vitx_handle h, void** graph_exec_out) {
  CAPI_TRY
  if (!h || !graph_exec_out) return fail(VITX_ERR_INVALID, "null argument");
  hipGraph_t graph = nullptr;
  hipError_t rc = hipStreamEndCapture(h->stream, &graph);
  if (rc != hipSuccess || !graph)
    return fail(VITX_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(rc) +
                                  " (run one eager step first: first-use allocations, kernel attributes and GEMM variant measurements cannot be captured)");
  hipGraphExec_t exec = nullptr;
  rc = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (rc != hipSuccess) return fail(VITX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(rc));
  *graph_exec_out = exec;
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_graph_launch(vitx_handle h, void* graph_exec) {
  CAPI_TRY
  if (!h || !graph_exec) return fail(VITX_ERR_INVALID, "null argument");
  CAPI_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, h->stream));
  return VITX_OK;
  CAPI_CATCH
}
int32_t vitx_graph_destroy(void* graph_exec) {
  CAPI_TRY
  if (graph_exec) CAPI_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_sync(vitx_handle h) {
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  CAPI_HIP(hipStreamSynchronize(h->stream));
  return VITX_OK;
}

int32_t vitx_set_grad_ready_callback(vitx_handle h, vitx_grad_ready_fn fn, void* user) {
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  h->grad_cb = fn;
  h->grad_cb_user = user;
  // a gradient-ready callback means collectives will run beside the backward pass: their workgroups take CUs away, and a
  // persistent GEMM (grid = #CUs, static tile lists) then waits for its late workgroups -- use one-tile-per-workgroup variants
  if (fn != nullptr) gemm_bf16_set_shared_gpu(1);   // sticky for the process
  return VITX_OK;
}

// ---- data parallel: the library's own RCCL exchange (comm.hip; librccl is dlopen'ed, libvitx does not link against it)
int32_t vitx_comm_unique_id(void* out128) {
  CAPI_TRY
  if (!out128) return fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  const int rc = comm_unique_id(out128, err);
  return rc == VITX_OK ? VITX_OK : fail(rc, err);
  CAPI_CATCH
}

int32_t vitx_comm_init(vitx_handle h, int32_t rank, int32_t world, const void* uid) {
  CAPI_TRY
  if (!h || !uid) return fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  const int rc = comm_init(h, rank, world, uid, err);
  return rc == VITX_OK ? VITX_OK : fail(rc, err);
  CAPI_CATCH
}

int32_t vitx_comm_overlap(vitx_handle h, int32_t enable, int64_t bucket_bytes, int32_t wire_bf16) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  std::string err;
  const int rc = comm_overlap(h, enable, bucket_bytes, wire_bf16, err);
  return rc == VITX_OK ? VITX_OK : fail(rc, err);
  CAPI_CATCH
}

int32_t vitx_comm_stats(vitx_handle h, int64_t* out4) {
  if (!h || !out4) return fail(VITX_ERR_INVALID, "null argument");
  comm_stats(h, out4);
  return VITX_OK;
}

int32_t vitx_comm_destroy(vitx_handle h) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  comm_destroy(h);
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_allreduce_grads(vitx_handle h) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  std::string err;
  const int rc = comm_finish(h, err);
  return rc == VITX_OK ? VITX_OK : fail(rc, err);
  CAPI_CATCH
}

int32_t vitx_debug_switches(char* out, int64_t cap, int64_t* needed) {
  CAPI_TRY
  int n = 0;
  const VitxEnvSwitch* t = vitx_env_table(&n);
  static const char* cls[] = {"tuning", "path", "diag"};
  std::string text;
  for (int i = 0; i < n; ++i) {
    const char* raw = getenv(t[i].name);      // (the one place that looks past vitx_env: to report a switch that is set but ignored)
    std::string state = "unset";
    if (raw) state = (t[i].cls == VITX_ENV_DIAG && !vitx_env_diag_build()) ? "ignored" : std::string("set=") + raw;
    text += std::string(t[i].name) + "\t" + cls[t[i].cls] + "\t" + state + "\t" + t[i].doc + "\n";
  }
  if (needed) *needed = (int64_t)text.size() + 1;
  if (out && cap > 0) {
    const size_t m = std::min<size_t>(text.size(), (size_t)cap - 1);
    std::memcpy(out, text.data(), m);
    out[m] = 0;
  }
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_profile_begin(vitx_handle h) {
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  h->prof_events.clear();
  h->profiling = true;
  return VITX_OK;
}

int32_t vitx_profile_end(vitx_handle h, vitx_kernel_stat* out, int32_t cap, int32_t* n_out) {
  CAPI_TRY
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  h->profiling = false;
  CAPI_HIP(hipStreamSynchronize(h->stream));
  std::vector<vitx_kernel_stat> stats(h->prof_names.size());
  for (size_t i = 0; i < stats.size(); ++i) {
    std::memset(&stats[i], 0, sizeof(vitx_kernel_stat));
    std::strncpy(stats[i].name, h->prof_names[i].c_str(), sizeof(stats[i].name) - 1);
  }
  for (auto& pe : h->prof_events) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, pe.e0, pe.e1);
    auto& s = stats[(size_t)pe.cls];
    s.launches += 1;
    s.total_ms += ms;
    s.flops += pe.flops;
    s.bytes += pe.bytes;
    if (pe.cls2 >= 0) {
      auto& s2 = stats[(size_t)pe.cls2];
      s2.launches += 1;
      s2.total_ms += ms;
      s2.flops += pe.flops;
      s2.bytes += pe.bytes;
    }
    (void)hipEventDestroy(pe.e0);
    (void)hipEventDestroy(pe.e1);
  }
  h->prof_events.clear();
  int n = 0;
  for (auto& s : stats) {
    if (s.launches == 0) continue;
    if (out && n < cap) out[n] = s;
    ++n;
  }
  if (n_out) *n_out = n;
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_workspace_bytes(vitx_handle h, int64_t* bytes) {
  if (!h) return fail(VITX_ERR_INVALID, "null handle");
  if (bytes) *bytes = h->ws_bytes;
  return VITX_OK;
}

int32_t vitx_debug_read(vitx_handle h, const char* which, int32_t layer, float* out_host, int64_t cap, int64_t* n_elems) {
  CAPI_TRY
  if (!h || !which) return fail(VITX_ERR_INVALID, "null argument");
  const vitx_config& c = h->cfg;
  const int b = h->last_b;
  // layer indexes the concatenation of all stages (CaiT: patch blocks then cls blocks)
  Stage* st = nullptr;
  int l = layer, nq = h->last_ntok, nk = h->last_ntok;
  for (auto& s : h->stages) {
    if (l < s.depth) { st = &s; break; }
    l -= s.depth;
  }
  const std::string w(which);
  const void* src = nullptr;
  int64_t rows = 0, cols = 0, ld = 0;
  bool is_t = false;
  if (w == "pooled_ln") { src = h->yh; rows = b; cols = ld = c.dim; is_t = true; }
  else if (w == "logits") { src = h->logits; rows = b; cols = c.num_classes; ld = h->nc_k; }
  else if (w == "dlogits") { src = h->dlogits; rows = b; cols = c.num_classes; ld = h->nc_k; }
  else if (w == "patches") { src = h->patches; rows = (int64_t)b * h->last_np; cols = h->pd; ld = h->pd_k; is_t = true; }
  else {
    if (!st) return fail(VITX_ERR_INVALID, "layer out of range");
    if (c.variant == VITX_VARIANT_CAIT) { nq = st->nq_max == 1 ? 1 : h->last_np; nk = st->nq_max == 1 ? 1 + h->last_np : h->last_np; }
    BlockActs& ba = st->ba[(size_t)l];
    rows = (int64_t)b * nq;
    if (w == "x_in" || w == "embed") { src = ba.x_in; cols = ld = c.dim; }
    else if (w == "x_mid") { src = ba.x_mid; cols = ld = c.dim; }
    else if (w == "x_out") { src = ba.x_out; cols = ld = c.dim; }
    else if (w == "y1") { src = ba.y1; cols = ld = c.dim; is_t = true; }
    else if (w == "y2") { src = ba.y2; cols = ld = c.dim; is_t = true; }
    else if (w == "qkv") { src = ba.qkv; cols = ld = 3 * h->inner; is_t = true; }
    else if (w == "q") { src = ba.q; cols = ld = h->inner; is_t = true; }
    else if (w == "kv") { src = ba.kv; rows = (int64_t)b * nk; cols = ld = 2 * h->inner; is_t = true; }
    else if (w == "attn_out") { src = ba.o; cols = ld = h->inner; is_t = true; }
    else if (w == "hpre") { src = ba.hpre; cols = ld = c.mlp_dim; is_t = true; }
    else if (w == "act") { src = ba.act; cols = ld = c.mlp_dim; is_t = true; }
    else return fail(VITX_ERR_INVALID, "unknown activation name");
  }
  if (!src) return fail(VITX_ERR_INVALID, "activation not available for this variant");
  const int64_t n = rows * cols;
  if (n_elems) *n_elems = n;
  if (!out_host) return VITX_OK;   // size query
  if (cap < n) return fail(VITX_ERR_INVALID, "output buffer too small");
  float* tmp = nullptr;
  CAPI_HIP(hipMalloc((void**)&tmp, (size_t)std::max<int64_t>(n, 1) * 4));
  launch_to_f32(src, is_t && h->bf16, ld, tmp, cols, (int)rows, (int)cols, h->stream);
  CAPI_HIP(hipMemcpyAsync(out_host, tmp, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(tmp);
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_check_gemm(vitx_handle h, int32_t kind, int32_t M, int32_t N, int32_t K, int32_t kernel, int32_t epilogue, float* errs2) {
  CAPI_TRY
  if (!h || !errs2) return fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  int rc = engine_check_gemm(h, kind, M, N, K, kernel, epilogue, errs2, err);
  if (rc != VITX_OK) return fail(rc, err);
  return VITX_OK;
  CAPI_CATCH
}

int32_t vitx_bench_gemm(vitx_handle h, int32_t M, int32_t N, int32_t K, int32_t kernel, int32_t epilogue, int32_t iters, float* avg_ms,
                        float* max_abs_err) {
  CAPI_TRY
  if (!h || !avg_ms) return fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  float me = -1.f;
  int rc = engine_bench_gemm(h, M, N, K, kernel, epilogue, iters, avg_ms, &me, err);
  if (max_abs_err) *max_abs_err = me;
  if (rc != VITX_OK) return fail(rc, err);
  return VITX_OK;
  CAPI_CATCH
}

}  // extern "C"
