// Knowledge distillation ("next" row f3): DistillMixin.call -- the student forward with a distillation
// token appended after the position embedding (distill.py:16-44) -- and DistillWrapper (distill.py:87-134): distill_mlp
// (LayerNormalization + Dense) on the split-off token, softmax cross-entropy against the labels, and the soft (temperature-scaled
// KL to the teacher) or hard (teacher argmax) distillation term.  The teacher is any model: its logits are an input
// (tf.stop_gradient, distill.py:114).  The student's blocks are the ordinary engine path at one more token.
#include <algorithm>
#include <cstring>

#include "engine.h"

int capi_fail(int code, const std::string& msg);   // capi.hip

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      err = std::string(#x) + ": " + hipGetErrorString(e_);                                         \
      return VITX_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)

struct vitx_distill {
  vitx_distill_config cfg{};
  vitx_engine* stu = nullptr;
  std::vector<ParamDesc> table;
  int64_t n_params = 0, n_arena = 0;
  float *params = nullptr, *grads = nullptr;
  int64_t tok = -1, ln_g = -1, ln_b = -1, w = -1, bias = -1;
  int d = 0, nc = 0, B = 0;
  std::vector<void*> allocs;
  float *img = nullptr, *labels = nullptr, *teacher = nullptr, *slog = nullptr, *dtok = nullptr, *yh = nullptr, *dlog = nullptr, *mean = nullptr,
        *rstd = nullptr;
  float *u_student = nullptr, *u_distill = nullptr;   // gradients for a unit cotangent, scaled at backward time
  float *ce = nullptr, *term = nullptr, *loss = nullptr, *coef = nullptr;
  float *g_slog = nullptr, *g_dlog = nullptr, *g_yh = nullptr, *g_dtok = nullptr, *ws = nullptr;
  bool have_fwd = false;
  int b = 0;
  float T = 1.f, alpha = 0.5f;
};

namespace {

constexpr float KERAS_EPS = 1e-7f;   // tf.keras.backend.epsilon(): the clip bounds inside keras.losses.KLDivergence

// One wave per image.  Writes ce[b] (distill.py:119), term[b] (per-image distillation term before the batch reduction / weighting)
// and the two logits gradients for unit cotangents.
//   soft, literal: term = sum_c y log(y / eps) with y = clip(softmax(teacher / T)): Keras' KLDivergence clips y_pred -- which is
//                  handed LOG-probabilities here (distill.py:122-124), all <= 0 -- to eps; constant in the student, gradient 0
//   soft, intended: term = KL(softmax(teacher / T) || softmax(distill / T));  d term / d distill = (softmax(distill / T) - y) / T
//   hard: term = -log_softmax(distill)[argmax teacher];  d term / d distill = softmax(distill) - onehot
__global__ __launch_bounds__(64) void distill_loss_kernel(const float* __restrict__ slog, const float* __restrict__ labels,
                                                          const float* __restrict__ dlog, const float* __restrict__ teacher, int b, int nc, float T,
                                                          int hard, int literal, float* __restrict__ ce, float* __restrict__ term,
                                                          float* __restrict__ u_student, float* __restrict__ u_distill) {
  const int row = blockIdx.x, lane = threadIdx.x;
  if (row >= b) return;
  const float* z = slog + (int64_t)row * nc;
  const float* y = labels + (int64_t)row * nc;
  const float* q = dlog + (int64_t)row * nc;
  const float* t = teacher + (int64_t)row * nc;
  // ---- student cross-entropy with (possibly soft) labels
  float m = -INFINITY;
  for (int c = lane; c < nc; c += 64) m = fmaxf(m, z[c]);
  m = wave_max(m);
  float s = 0.f, ysum = 0.f, yz = 0.f;
  for (int c = lane; c < nc; c += 64) { s += expf(z[c] - m); ysum += y[c]; yz += y[c] * z[c]; }
  s = wave_sum(s); ysum = wave_sum(ysum); yz = wave_sum(yz);
  const float lse = m + logf(s);
  if (lane == 0) ce[row] = ysum * lse - yz;                       // -sum_c y_c (z_c - lse)
  for (int c = lane; c < nc; c += 64) u_student[(int64_t)row * nc + c] = expf(z[c] - lse) * ysum - y[c];
  // ---- distillation term
  const float invT = hard ? 1.f : 1.f / T;
  float mq = -INFINITY, mt = -INFINITY;
  int arg = 0;
  for (int c = lane; c < nc; c += 64) {
    mq = fmaxf(mq, q[c] * invT);
    if (t[c] * invT > mt) { mt = t[c] * invT; arg = c; }           // first maximum within the lane's strided subsequence
  }
  // argmax over the row: larger value wins, ties go to the smaller index (tf.argmax)
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(mt, off, 64);
    const int oa = __shfl_xor(arg, off, 64);
    if (om > mt || (om == mt && oa < arg)) { mt = om; arg = oa; }
  }
  mq = wave_max(mq);
  float sq = 0.f, st = 0.f;
  for (int c = lane; c < nc; c += 64) { sq += expf(q[c] * invT - mq); st += expf(t[c] * invT - mt); }
  sq = wave_sum(sq); st = wave_sum(st);
  const float lse_q = mq + logf(sq), lse_t = mt + logf(st);
  float acc = 0.f;
  for (int c = lane; c < nc; c += 64) {
    const float lq = q[c] * invT - lse_q;                          // log_softmax(distill / T)
    float g;
    if (hard) {
      g = expf(lq) - (c == arg ? 1.f : 0.f);
      if (c == arg) acc -= lq;
    } else {
      const float yt = expf(t[c] * invT - lse_t);                  // softmax(teacher / T)
      if (literal) {
        const float yc = fminf(fmaxf(yt, KERAS_EPS), 1.f);
        acc += yc * logf(yc / KERAS_EPS);
        g = 0.f;
      } else {
        if (yt > 0.f) acc += yt * (logf(yt) - lq);
        g = (expf(lq) - yt) * invT;
      }
    }
    u_distill[(int64_t)row * nc + c] = g;
  }
  acc = wave_sum(acc);
  if (lane == 0) term[row] = acc;
}

// loss[b] = ce[b] (1 - alpha) + alpha * (hard ? term[b] : T^2 * sum(term) / batch)      (distill.py:126-134)
__global__ __launch_bounds__(64) void distill_combine_kernel(const float* __restrict__ ce, const float* __restrict__ term, int b, float alpha, float T,
                                                             int hard, float* __restrict__ loss) {
  float s = 0.f;
  if (!hard) {
    for (int i = threadIdx.x; i < b; i += 64) s += term[i];
    s = wave_sum(s);
    s = __shfl(s, 0, 64) / (float)b * T * T;
  }
  for (int i = threadIdx.x; i < b; i += 64) loss[i] = ce[i] * (1.f - alpha) + (hard ? term[i] : s) * alpha;
}

// out[row, :] = unit[row, :] * coef[row]
__global__ void scale_rows_kernel(const float* __restrict__ unit, const float* __restrict__ coef, int64_t rows, int nc, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * nc) return;
  out[e] = unit[e] * coef[e / nc];
}

int64_t add_param(vitx_distill* m, const std::string& name, std::vector<int64_t> shape) {
  ParamDesc p;
  p.name = name; p.shape = shape; p.count = 1;
  for (int64_t s : shape) p.count *= s;
  p.offset = m->n_params; p.aoff = m->n_arena;
  m->n_params += p.count;
  m->n_arena += round_up(p.count, 4);
  m->table.push_back(p);
  return p.aoff;
}

int dist_alloc(vitx_distill* m, void** p, size_t bytes, std::string& err) {
  bytes = (size_t)round_up((int64_t)std::max<size_t>(bytes, 16), 256);
  HIPCHK(hipMalloc(p, bytes));
  HIPCHK(hipMemsetAsync(*p, 0, bytes, m->stu->stream));
  m->allocs.push_back(*p);
  return VITX_OK;
}
#define DMALLOC(ptr, bytes)                                         \
  do {                                                              \
    int rc_ = dist_alloc(m, (void**)&(ptr), (size_t)(bytes), err);  \
    if (rc_ != VITX_OK) return rc_;                                 \
  } while (0)

bool al16(std::initializer_list<const void*> ps) {
  for (const void* p : ps) if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

int distill_create(vitx_engine* stu, const vitx_distill_config& cfg, vitx_distill** out, std::string& err) {
  if (stu->cfg.variant == VITX_VARIANT_CAIT || stu->cfg.variant == VITX_VARIANT_PATCH_MERGER) { err = "student must be a vision transformer"; return VITX_ERR_INVALID; }   // distill.py:91 (Distillable* classes)
  if (!(cfg.temperature > 0.f)) { err = "temperature must be positive"; return VITX_ERR_INVALID; }
  vitx_distill* m = new vitx_distill();
  m->cfg = cfg; m->stu = stu;
  m->d = stu->cfg.dim; m->nc = stu->cfg.num_classes; m->B = stu->cfg.max_batch;
  // attribute order of DistillWrapper.__init__ (distill.py:101-106)
  m->tok = add_param(m, "distillation_token", {1, 1, m->d});
  m->ln_g = add_param(m, "distill_mlp.norm.gamma", {m->d});
  m->ln_b = add_param(m, "distill_mlp.norm.beta", {m->d});
  m->w = add_param(m, "distill_mlp.kernel", {m->d, m->nc});
  m->bias = add_param(m, "distill_mlp.bias", {m->nc});
  const int64_t B = m->B, d = m->d, nc = m->nc;
  DMALLOC(m->params, (size_t)m->n_arena * 4);
  DMALLOC(m->grads, (size_t)m->n_arena * 4);
  DMALLOC(m->img, (size_t)B * stu->cfg.image_h * stu->cfg.image_w * stu->cfg.channels * 4);
  DMALLOC(m->labels, (size_t)B * nc * 4); DMALLOC(m->teacher, (size_t)B * nc * 4); DMALLOC(m->slog, (size_t)B * nc * 4);
  DMALLOC(m->dlog, (size_t)B * nc * 4); DMALLOC(m->u_student, (size_t)B * nc * 4); DMALLOC(m->u_distill, (size_t)B * nc * 4);
  DMALLOC(m->g_slog, (size_t)B * nc * 4); DMALLOC(m->g_dlog, (size_t)B * nc * 4);
  DMALLOC(m->dtok, (size_t)B * d * 4); DMALLOC(m->yh, (size_t)B * d * 4); DMALLOC(m->g_yh, (size_t)B * d * 4); DMALLOC(m->g_dtok, (size_t)B * d * 4);
  DMALLOC(m->mean, (size_t)B * 4); DMALLOC(m->rstd, (size_t)B * 4); DMALLOC(m->ce, (size_t)B * 4); DMALLOC(m->term, (size_t)B * 4);
  DMALLOC(m->loss, (size_t)B * 4); DMALLOC(m->coef, (size_t)B * 2 * 4);
  DMALLOC(m->ws, (size_t)(std::max<int64_t>(layernorm_bwd_ws_elems((int)d), colsum_ws_elems((int)std::max(nc, d))) + 64) * 4);
  HIPCHK(hipStreamSynchronize(stu->stream));
  *out = m;
  return VITX_OK;
}

void distill_destroy(vitx_distill* m) {
  if (!m) return;
  (void)hipDeviceSynchronize();
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
}

// DistillWrapper.call (distill.py:107-134) with the teacher's logits already computed
int distill_forward(vitx_distill* m, const float* img_dev, const float* labels_dev, const float* teacher_dev, int b, int H, int W, int training,
                    uint64_t seed, float temperature, float alpha, std::string& err) {
  vitx_engine* e = m->stu;
  hipStream_t s = e->stream;
  const int d = m->d, nc = m->nc;
  m->have_fwd = false;
  m->T = temperature > 0.f ? temperature : m->cfg.temperature;        // distill.py:111
  m->alpha = alpha >= 0.f ? alpha : m->cfg.alpha;                     // distill.py:110
  const float* P = m->params;
  int rc;
  if ((rc = engine_forward(e, img_dev, b, H, W, training, seed, m->slog, err, P + m->tok, m->dtok)) != VITX_OK) return rc;   // distill.py:116
  launch_layernorm_fwd(m->dtok, d, P + m->ln_g, P + m->ln_b, m->yh, 0, d, m->mean, m->rstd, b, d, e->cfg.ln_eps, s);     // distill.py:104
  {                                                                                                                   // distill.py:105
    GenericGemmArgs g;
    g.A = m->yh; g.B = P + m->w; g.M = b; g.N = nc; g.K = d; g.sam = d; g.sak = 1; g.sbk = nc; g.sbn = 1;
    EpiParams ep;
    ep.out = m->dlog; ep.ldo = nc; ep.M = b; ep.N = nc; ep.bias = P + m->bias;
    ep.vec_ok = (nc % 4 == 0) && al16({m->dlog, ep.bias});
    launch_gemm_generic(g, ep, EPI_STORE_F32, 0, 0, 0, s);
  }
  hipLaunchKernelGGL(distill_loss_kernel, dim3((unsigned)b), dim3(64), 0, s, m->slog, labels_dev, m->dlog, teacher_dev, b, nc, m->T, m->cfg.hard,
                     m->cfg.literal_loss, m->ce, m->term, m->u_student, m->u_distill);
  hipLaunchKernelGGL(distill_combine_kernel, dim3(1), dim3(64), 0, s, m->ce, m->term, b, m->alpha, m->T, m->cfg.hard, m->loss);
  m->have_fwd = true; m->b = b;
  return VITX_OK;
}

// dloss_host [b] or null (= ones: tf's tape.gradient of a non-scalar target differentiates its sum)
int distill_backward(vitx_distill* m, const float* dloss_host, std::string& err, float* dinput_dev = nullptr) {
  if (!m->have_fwd) { err = "backward requires a preceding forward"; return VITX_ERR_STATE; }
  vitx_engine* e = m->stu;
  hipStream_t s = e->stream;
  const int b = m->b, d = m->d, nc = m->nc;
  const float* P = m->params;
  float* G = m->grads;
  std::vector<float> coef((size_t)2 * b);
  double S = 0.0;
  for (int i = 0; i < b; ++i) S += dloss_host ? dloss_host[i] : 1.0;
  for (int i = 0; i < b; ++i) {
    const float dl = dloss_host ? dloss_host[i] : 1.f;
    coef[(size_t)i] = dl * (1.f - m->alpha);
    coef[(size_t)b + i] = m->cfg.hard ? dl * m->alpha : (float)(S * m->alpha * m->T * m->T / b);   // the soft term is one scalar shared by every entry
  }
  HIPCHK(hipMemcpyAsync(m->coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));   // coef is a stack-lifetime host buffer
  const int64_t n = (int64_t)b * nc;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, m->u_student, m->coef, (int64_t)b, nc, m->g_slog);
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, m->u_distill, m->coef + b, (int64_t)b, nc, m->g_dlog);
  launch_fill_zero(G, m->n_arena * 4, s);
  {  // distill_mlp Dense: d yh = dL W^T, dW = yh^T dL, db = column sums
    GenericGemmArgs g;
    g.A = m->g_dlog; g.B = P + m->w; g.M = b; g.N = d; g.K = nc; g.sam = nc; g.sak = 1; g.sbk = 1; g.sbn = nc;
    EpiParams ep;
    ep.out = m->g_yh; ep.ldo = d; ep.M = b; ep.N = d; ep.vec_ok = (d % 4 == 0) && al16({m->g_yh});
    launch_gemm_generic(g, ep, EPI_STORE_F32, 0, 0, 0, s);
    GenericGemmArgs w;
    w.A = m->yh; w.B = m->g_dlog; w.M = d; w.N = nc; w.K = b; w.sam = 1; w.sak = d; w.sbk = nc; w.sbn = 1;
    EpiParams ew;
    ew.out = G + m->w; ew.ldo = nc; ew.M = d; ew.N = nc; ew.vec_ok = (nc % 4 == 0) && al16({ew.out});
    launch_gemm_generic(w, ew, EPI_STORE_F32, 0, 0, 0, s);
    launch_colsum(m->g_dlog, 0, nc, b, nc, m->ws, G + m->bias, s);
  }
  launch_layernorm_bwd(m->g_yh, 0, d, m->dtok, d, m->mean, m->rstd, P + m->ln_g, nullptr, 0, m->g_dtok, d, nullptr, 0, m->ws, G + m->ln_g, G + m->ln_b,
                       nullptr, b, d, s);
  return engine_backward(e, m->g_slog, dinput_dev, err, m->g_dtok, G + m->tok);
}

}  // namespace

#define D_TRY try {
#define D_CATCH                                                                       \
  }                                                                                   \
  catch (const std::exception& ex) { return capi_fail(VITX_ERR_INVALID, ex.what()); } \
  catch (...) { return capi_fail(VITX_ERR_INVALID, "unknown C++ exception"); }
#define D_HIP(x)                                                                                           \
  do {                                                                                                     \
    hipError_t e_ = (x);                                                                                   \
    if (e_ != hipSuccess) return capi_fail(VITX_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));  \
  } while (0)

static float* distill_scratch(vitx_engine* h) {
  if (!h->distill_ws) {
    void* p = nullptr;
    if (hipMalloc(&p, ((size_t)h->cfg.max_batch + 1) * h->cfg.dim * 4 + 256) != hipSuccess) return nullptr;
    h->allocs.push_back(p);
    h->distill_ws = (float*)p;
  }
  return h->distill_ws;
}

extern "C" {

// ---- DistillableViT.call(img, distill_token) (distill.py:16-44) and its VJP on host buffers
int32_t vitx_forward_distill(vitx_handle h, const float* img_host, int32_t b, int32_t H, int32_t W, int32_t training, uint64_t seed,
                             const float* distill_token_host, float* logits_host, float* distill_tokens_host) {
  D_TRY
  const int np_in = h ? h->next_patch_np : 0;   // > 0: img_host holds patch rows [b, np, patch_dim] (vitx_set_patch_input, one shot: consumed
  if (h) h->next_patch_np = 0;                  //   before any validation can fail, so that a rejected call does not leave the handle armed)
  if (!h || !img_host || !distill_token_host || !logits_host || !distill_tokens_host) return capi_fail(VITX_ERR_INVALID, "null argument");
  if (b <= 0 || b > h->cfg.max_batch) return capi_fail(VITX_ERR_INVALID, "batch must be in [1, max_batch]");
  if (!np_in && (H <= 0 || W <= 0 || H > h->cfg.image_h || W > h->cfg.image_w)) return capi_fail(VITX_ERR_INVALID, "image larger than the configured image_size");
  const int d = h->cfg.dim, nc = h->cfg.num_classes;
  hipStream_t s = h->stream;
  float* tok = distill_scratch(h);   // row 0 = the token, rows 1.. = the returned per-image tokens
  if (!tok) return capi_fail(VITX_ERR_HIP, "hipMalloc failed");
  const size_t in_elems = np_in ? (size_t)b * np_in * h->pd : (size_t)b * H * W * h->cfg.channels;
  D_HIP(hipMemcpyAsync(h->img_dev, img_host, in_elems * 4, hipMemcpyHostToDevice, s));
  D_HIP(hipMemcpyAsync(tok, distill_token_host, (size_t)d * 4, hipMemcpyHostToDevice, s));
  std::string err;
  if (np_in) { h->fwd_patches = h->img_dev; h->fwd_np = np_in; }   // set right before the call that consumes it: no error exit in between
  int rc = engine_forward(h, h->img_dev, b, H, W, training, seed, nullptr, err, tok, tok + d);
  h->fwd_patches = nullptr;
  if (rc != VITX_OK) return capi_fail(rc, err);
  D_HIP(hipMemcpy2DAsync(logits_host, (size_t)nc * 4, h->logits, (size_t)h->nc_k * 4, (size_t)nc * 4, (size_t)b, hipMemcpyDeviceToHost, s));
  D_HIP(hipMemcpyAsync(distill_tokens_host, tok + d, (size_t)b * d * 4, hipMemcpyDeviceToHost, s));
  D_HIP(hipStreamSynchronize(s));
  return VITX_OK;
  D_CATCH
}
int32_t vitx_backward_distill(vitx_handle h, const float* dlogits_host, const float* d_distill_tokens_host, float* d_distill_token_host,
                              float* dimg_host_or_null) {
  D_TRY
  if (!h || !dlogits_host) return capi_fail(VITX_ERR_INVALID, "null argument");
  if (!h->have_fwd || !h->last_extra) return capi_fail(VITX_ERR_STATE, "backward_distill requires a preceding forward_distill");
  const int b = h->last_b, nc = h->cfg.num_classes, d = h->cfg.dim;
  hipStream_t s = h->stream;
  float* tok = distill_scratch(h);
  if (!tok) return capi_fail(VITX_ERR_HIP, "hipMalloc failed");
  D_HIP(hipMemcpy2DAsync(h->dlogits, (size_t)h->nc_k * 4, dlogits_host, (size_t)nc * 4, (size_t)nc * 4, (size_t)b, hipMemcpyHostToDevice, s));
  if (d_distill_tokens_host) D_HIP(hipMemcpyAsync(tok + d, d_distill_tokens_host, (size_t)b * d * 4, hipMemcpyHostToDevice, s));
  float* dimg_dev = dimg_host_or_null ? h->img_dev : nullptr;
  std::string err;
  float* dd = d_distill_tokens_host ? tok + d : nullptr;
  float* dt = tok;
  int rc = engine_backward(h, nullptr, dimg_dev, err, dd, dt);
  if (rc != VITX_OK) return capi_fail(rc, err);
  if (d_distill_token_host) D_HIP(hipMemcpyAsync(d_distill_token_host, dt, (size_t)d * 4, hipMemcpyDeviceToHost, s));
  if (dimg_host_or_null)
    D_HIP(hipMemcpyAsync(dimg_host_or_null, dimg_dev, (size_t)b * h->last_H * h->last_W * h->cfg.channels * 4, hipMemcpyDeviceToHost, s));
  D_HIP(hipStreamSynchronize(s));
  return VITX_OK;
  D_CATCH
}

// ---- DistillWrapper (distill.py:87-134)
int32_t vitx_distill_create(vitx_handle student, const vitx_distill_config* cfg, vitx_distill_handle* out) {
  D_TRY
  if (!student || !cfg || !out) return capi_fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  vitx_distill* m = nullptr;
  int rc = distill_create(student, *cfg, &m, err);
  if (rc != VITX_OK) return capi_fail(rc, err);
  *out = m;
  return VITX_OK;
  D_CATCH
}
int32_t vitx_distill_destroy(vitx_distill_handle m) {
  D_TRY
  distill_destroy(m);
  return VITX_OK;
  D_CATCH
}
int32_t vitx_distill_param_table_size(vitx_distill_handle m, int64_t* n_tensors, int64_t* n_elems) {
  if (!m) return capi_fail(VITX_ERR_INVALID, "null handle");
  if (n_tensors) *n_tensors = (int64_t)m->table.size();
  if (n_elems) *n_elems = m->n_params;
  return VITX_OK;
}
int32_t vitx_distill_param_table_entry(vitx_distill_handle m, int64_t index, char* name, int32_t name_cap, int64_t shape[4], int32_t* rank,
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
static int dist_copy_blob(vitx_distill* m, float* arena, float* host, int64_t n, bool to_device) {
  if (n != m->n_params) return capi_fail(VITX_ERR_INVALID, "blob size does not match the wrapper's parameter table");
  hipStream_t s = m->stu->stream;
  for (auto& p : m->table) {
    if (to_device) D_HIP(hipMemcpyAsync(arena + p.aoff, host + p.offset, (size_t)p.count * 4, hipMemcpyHostToDevice, s));
    else D_HIP(hipMemcpyAsync(host + p.offset, arena + p.aoff, (size_t)p.count * 4, hipMemcpyDeviceToHost, s));
  }
  D_HIP(hipStreamSynchronize(s));
  return VITX_OK;
}
int32_t vitx_distill_set_params(vitx_distill_handle m, const float* host_blob, int64_t n) {
  D_TRY
  if (!m || !host_blob) return capi_fail(VITX_ERR_INVALID, "null argument");
  return dist_copy_blob(m, m->params, const_cast<float*>(host_blob), n, true);
  D_CATCH
}
int32_t vitx_distill_get_params(vitx_distill_handle m, float* host_blob, int64_t n) {
  D_TRY
  if (!m || !host_blob) return capi_fail(VITX_ERR_INVALID, "null argument");
  return dist_copy_blob(m, m->params, host_blob, n, false);
  D_CATCH
}
int32_t vitx_distill_get_grads(vitx_distill_handle m, float* host_blob, int64_t n) {
  D_TRY
  if (!m || !host_blob) return capi_fail(VITX_ERR_INVALID, "null argument");
  return dist_copy_blob(m, m->grads, host_blob, n, false);
  D_CATCH
}

int32_t vitx_distill_forward(vitx_distill_handle m, const float* img_host, const float* labels_host, const float* teacher_logits_host, int32_t b,
                             int32_t H, int32_t W, int32_t training, uint64_t seed, float temperature, float alpha, float* loss_host) {
  D_TRY
  const int np_in = (m && m->stu) ? m->stu->next_patch_np : 0;   // > 0: img_host holds patch rows [b, np, patch_dim] (vitx_set_patch_input on the
  if (m && m->stu) m->stu->next_patch_np = 0;                    //   student, one shot: consumed before any validation can fail)
  if (!m || !img_host || !labels_host || !teacher_logits_host) return capi_fail(VITX_ERR_INVALID, "null argument");
  const vitx_config& c = m->stu->cfg;
  if (b <= 0 || b > c.max_batch) return capi_fail(VITX_ERR_INVALID, "batch must be in [1, max_batch]");
  if (!np_in && (H <= 0 || W <= 0 || H > c.image_h || W > c.image_w)) return capi_fail(VITX_ERR_INVALID, "image larger than the configured image_size");
  hipStream_t s = m->stu->stream;
  const size_t in_elems = np_in ? (size_t)b * np_in * m->stu->pd : (size_t)b * H * W * c.channels;
  D_HIP(hipMemcpyAsync(m->img, img_host, in_elems * 4, hipMemcpyHostToDevice, s));
  D_HIP(hipMemcpyAsync(m->labels, labels_host, (size_t)b * m->nc * 4, hipMemcpyHostToDevice, s));
  D_HIP(hipMemcpyAsync(m->teacher, teacher_logits_host, (size_t)b * m->nc * 4, hipMemcpyHostToDevice, s));
  std::string err;
  if (np_in) { m->stu->fwd_patches = m->img; m->stu->fwd_np = np_in; }   // set right before the call that consumes it
  int rc = distill_forward(m, m->img, m->labels, m->teacher, b, H, W, training, seed, temperature, alpha, err);
  m->stu->fwd_patches = nullptr;                                          // (also when distill_forward failed before reaching the engine)
  if (rc != VITX_OK) return capi_fail(rc, err);
  if (loss_host) D_HIP(hipMemcpyAsync(loss_host, m->loss, (size_t)b * 4, hipMemcpyDeviceToHost, s));
  D_HIP(hipStreamSynchronize(s));
  return VITX_OK;
  D_CATCH
}
int32_t vitx_distill_forward_dev(vitx_distill_handle m, const float* img_dev, const float* labels_dev, const float* teacher_logits_dev, int32_t b,
                                 int32_t H, int32_t W, int32_t training, uint64_t seed, float temperature, float alpha, float* loss_dev_or_null) {
  D_TRY
  if (!m || !img_dev || !labels_dev || !teacher_logits_dev) return capi_fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  int rc = distill_forward(m, img_dev, labels_dev, teacher_logits_dev, b, H, W, training, seed, temperature, alpha, err);
  if (rc != VITX_OK) return capi_fail(rc, err);
  if (loss_dev_or_null) D_HIP(hipMemcpyAsync(loss_dev_or_null, m->loss, (size_t)b * 4, hipMemcpyDeviceToDevice, m->stu->stream));
  return VITX_OK;
  D_CATCH
}
int32_t vitx_distill_backward(vitx_distill_handle m, const float* dloss_host_or_null) {
  D_TRY
  if (!m) return capi_fail(VITX_ERR_INVALID, "null handle");
  std::string err;
  int rc = distill_backward(m, dloss_host_or_null, err);
  if (rc != VITX_OK) return capi_fail(rc, err);
  return VITX_OK;
  D_CATCH
}
// the same, also returning d(loss)/d(input) of the student's forward: d(img) [b, H, W, C], or d(patches) [b, np, patch_dim] when that
// forward took patch rows (a T2T-ViT student: the tokenizer in front of the handle needs it to continue the chain)
int32_t vitx_distill_backward_input(vitx_distill_handle m, const float* dloss_host_or_null, float* dinput_host) {
  D_TRY
  if (!m || !dinput_host) return capi_fail(VITX_ERR_INVALID, "null argument");
  std::string err;
  int rc = distill_backward(m, dloss_host_or_null, err, m->img);
  if (rc != VITX_OK) return capi_fail(rc, err);
  vitx_engine* e = m->stu;
  D_HIP(hipMemcpyAsync(dinput_host, m->img, (size_t)e->last_b * e->last_H * e->last_W * e->cfg.channels * 4, hipMemcpyDeviceToHost, e->stream));
  D_HIP(hipStreamSynchronize(e->stream));
  return VITX_OK;
  D_CATCH
}
int32_t vitx_distill_read(vitx_distill_handle m, const char* which, float* out_host, int64_t cap, int64_t* n_elems) {
  D_TRY
  if (!m || !which || !out_host) return capi_fail(VITX_ERR_INVALID, "null argument");
  if (!m->have_fwd) return capi_fail(VITX_ERR_STATE, "read requires a preceding forward");
  const std::string w = which;
  const float* src = nullptr;
  int64_t n = 0;
  if (w == "student_logits") { src = m->slog; n = (int64_t)m->b * m->nc; }
  else if (w == "distill_logits") { src = m->dlog; n = (int64_t)m->b * m->nc; }
  else if (w == "distill_tokens") { src = m->dtok; n = (int64_t)m->b * m->d; }
  else return capi_fail(VITX_ERR_INVALID, "unknown tensor name");
  if (n_elems) *n_elems = n;
  if (n > cap) return capi_fail(VITX_ERR_INVALID, "output buffer too small");
  D_HIP(hipMemcpyAsync(out_host, src, (size_t)n * 4, hipMemcpyDeviceToHost, m->stu->stream));
  D_HIP(hipStreamSynchronize(m->stu->stream));
  return VITX_OK;
  D_CATCH
}

}  // extern "C"
