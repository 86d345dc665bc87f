// Engine: parameter table, memory plan and launch sequences (see engine.h).
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      err = std::string(#x) + ": " + hipGetErrorString(e_);                                         \
      return VITX_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)

// ------------------------------------------------------------------------------------------------
// parameter table
// ------------------------------------------------------------------------------------------------
std::string build_param_table(const vitx_config& c, std::vector<ParamDesc>& out) {
  out.clear();
  if (c.variant < 0 || c.variant > 3) return "unknown variant";
  if (c.patch_h <= 0 || c.patch_w <= 0 || c.image_h <= 0 || c.image_w <= 0) return "image/patch size must be positive";
  if (c.image_h % c.patch_h != 0 || c.image_w % c.patch_w != 0)
    return "Image dimensions must be divisible by the patch size.";   // vit.py:136, deepvit.py:117, cait.py:160
  if (c.variant != VITX_VARIANT_CAIT && c.pool != VITX_POOL_CLS && c.pool != VITX_POOL_MEAN)
    return "pool type must be either cls (cls token) or mean (mean pooling)";  // vit.py:139
  if (c.dim <= 0 || c.depth < 0 || c.heads <= 0 || c.dim_head <= 0 || c.mlp_dim <= 0 || c.num_classes <= 0 || c.channels <= 0)
    return "dim/depth/heads/dim_head/mlp_dim/num_classes must be positive";
  const int64_t d = c.dim, h = c.heads, dh = c.dim_head, m = c.mlp_dim, nc = c.num_classes, inner = h * dh;
  const int64_t np = (int64_t)(c.image_h / c.patch_h) * (c.image_w / c.patch_w);
  const int64_t pd = (int64_t)c.patch_h * c.patch_w * c.channels;
  int64_t off = 0, aoff = 0;
  auto add = [&](const std::string& n, std::vector<int64_t> s) {
    ParamDesc p;
    p.name = n;
    p.shape = s;
    p.count = 1;
    for (auto v : s) p.count *= v;
    p.offset = off;
    p.aoff = aoff;
    off += p.count;
    aoff += round_up(p.count, 4);
    out.push_back(p);
  };
  const bool cait = c.variant == VITX_VARIANT_CAIT, deep = c.variant == VITX_VARIANT_DEEPVIT, merger = c.variant == VITX_VARIANT_PATCH_MERGER;
  if (merger && c.patch_merge_num_tokens <= 0) return "patch_merge_num_tokens must be positive";
  add("pos_embedding", {1, cait ? np : np + 1, d});
  if (!merger) add("cls_token", {1, 1, d});          // vit_with_patch_merger.ViT has no cls token (vit_with_patch_merger.py:163-166)
  add("patch_embedding.kernel", {pd, d});
  add("patch_embedding.bias", {d});
  auto block = [&](const std::string& pre) {
    if (cait) add(pre + ".attn.scale", {1, 1, d});
    add(pre + ".attn.norm.gamma", {d});
    add(pre + ".attn.norm.beta", {d});
    if (cait) {
      add(pre + ".attn.to_q.kernel", {d, inner});
      add(pre + ".attn.to_kv.kernel", {d, 2 * inner});
      add(pre + ".attn.mix_heads_pre_attn", {h, h});
      add(pre + ".attn.mix_heads_post_attn", {h, h});
    } else {
      add(pre + ".attn.to_qkv.kernel", {d, 3 * inner});
    }
    if (deep) {
      add(pre + ".attn.reattn_weights", {h, h});
      add(pre + ".attn.reattn_norm.gamma", {h});
      add(pre + ".attn.reattn_norm.beta", {h});
    }
    const bool project_out = !((c.variant == VITX_VARIANT_VIT || merger) && h == 1 && dh == d);  // vit.py:53
    if (project_out) {
      add(pre + ".attn.to_out.kernel", {inner, d});
      add(pre + ".attn.to_out.bias", {d});
    }
    if (cait) add(pre + ".mlp.scale", {1, 1, d});
    add(pre + ".mlp.norm.gamma", {d});
    add(pre + ".mlp.norm.beta", {d});
    add(pre + ".mlp.fc1.kernel", {d, m});
    add(pre + ".mlp.fc1.bias", {m});
    add(pre + ".mlp.fc2.kernel", {m, d});
    add(pre + ".mlp.fc2.bias", {d});
  };
  if (merger) {   // attribute order of its Transformer: patch_merger before the layers (vit_with_patch_merger.py:118-124)
    add("transformer.patch_merger.norm.gamma", {d});
    add("transformer.patch_merger.norm.beta", {d});
    add("transformer.patch_merger.queries", {c.patch_merge_num_tokens, d});
  }
  const int P = c.num_parallel_branches > 1 ? c.num_parallel_branches : 1;
  if (P > 1 && (c.variant != VITX_VARIANT_VIT || P > 8)) return "num_parallel_branches needs the ViT variant and at most 8 branches";
  if (cait) {
    for (int i = 0; i < c.depth; ++i) block("patch_transformer." + std::to_string(i));
    for (int i = 0; i < c.cls_depth; ++i) block("cls_transformer." + std::to_string(i));
  } else if (P > 1) {
    // parallel_vit.py:104-111: layers[l] = [Parallel([PreNorm(Attention)] * P), Parallel([PreNorm(MLP)] * P)]
    const bool project_out = !(h == 1 && dh == d);
    for (int l = 0; l < c.depth; ++l) {
      for (int i = 0; i < P; ++i) {
        const std::string pre = "transformer." + std::to_string(l) + ".attn." + std::to_string(i);
        add(pre + ".norm.gamma", {d});
        add(pre + ".norm.beta", {d});
        add(pre + ".to_qkv.kernel", {d, 3 * inner});
        if (project_out) { add(pre + ".to_out.kernel", {inner, d}); add(pre + ".to_out.bias", {d}); }
      }
      for (int i = 0; i < P; ++i) {
        const std::string pre = "transformer." + std::to_string(l) + ".mlp." + std::to_string(i);
        add(pre + ".norm.gamma", {d});
        add(pre + ".norm.beta", {d});
        add(pre + ".fc1.kernel", {d, m});
        add(pre + ".fc1.bias", {m});
        add(pre + ".fc2.kernel", {m, d});
        add(pre + ".fc2.bias", {d});
      }
    }
  } else {
    for (int i = 0; i < c.depth; ++i) block("transformer." + std::to_string(i));
  }
  add("mlp_head.norm.gamma", {d});
  add("mlp_head.norm.beta", {d});
  add("mlp_head.kernel", {d, nc});
  add("mlp_head.bias", {nc});
  return "";
}

// ------------------------------------------------------------------------------------------------
// profiling scope
// ------------------------------------------------------------------------------------------------
int prof_class(vitx_engine* e, const char* name) {
  for (size_t i = 0; i < e->prof_names.size(); ++i)
    if (e->prof_names[i] == name) return (int)i;
  e->prof_names.push_back(name);
  return (int)e->prof_names.size() - 1;
}
struct Prof {
  vitx_engine* e;
  ProfEvent pe;
  bool on;
  hipStream_t st;
  Prof(vitx_engine* e_, const char* name, double flops, double bytes, const char* name2 = nullptr, hipStream_t st_ = nullptr)
      : e(e_), on(e_->profiling), st(st_ ? st_ : e_->stream) {
    if (!on) return;
    pe.cls = prof_class(e, name);
    pe.cls2 = name2 ? prof_class(e, name2) : -1;
    pe.flops = flops;
    pe.bytes = bytes;
    (void)hipEventCreate(&pe.e0);
    (void)hipEventCreate(&pe.e1);
    (void)hipEventRecord(pe.e0, st);
  }
  ~Prof() {
    if (!on) return;
    (void)hipEventRecord(pe.e1, st);
    e->prof_events.push_back(pe);
  }
};

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
static inline char* boff(void* p, int64_t elems, int esz) { return (char*)p + elems * esz; }
static inline const char* boff(const void* p, int64_t elems, int esz) { return (const char*)p + elems * esz; }

static void finalize_epi(EpiParams& ep) {
  auto al = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
  ep.vec_ok = (ep.ldo % 4 == 0) && (ep.ldo2 % 4 == 0) && (ep.ldr % 4 == 0) && (ep.ldaux % 4 == 0) && al(ep.out) && al(ep.out2) &&
              al(ep.bias) && al(ep.resid) && al(ep.scale) && al(ep.aux) && al(ep.pos) && (ep.out_batch_stride % 4 == 0) &&
              (ep.out_head_stride % 4 == 0) && (ep.partial_stride % 4 == 0);
  static const int wide_env = [] { const char* v = vitx_env("VITX_EPI_WIDE"); return v ? atoi(v) : 1; }();
  ep.wide_ok = wide_env && ep.vec_ok && (ep.ldo % 8 == 0) && (ep.ldo2 % 8 == 0) && (ep.ldaux % 8 == 0);
}

static int dalloc(vitx_engine* e, void** p, size_t bytes, bool t_buffer, std::string& err) {
  if (bytes == 0) bytes = 16;
  bytes = (size_t)round_up((int64_t)bytes, 256) + 8192;   // slack: GEMM tiles may read (never use) a little past a buffer's last row
  HIPCHK(hipMalloc(p, bytes));
  HIPCHK(hipMemsetAsync(*p, 0, bytes, e->stream));
  e->allocs.push_back(*p);
  if (t_buffer) e->t_buffers.push_back({*p, bytes});
  e->ws_bytes += (int64_t)bytes;
  return VITX_OK;
}
#define DALLOC(ptr, bytes, tb)                                             \
  do {                                                                     \
    int rc_ = dalloc(e, (void**)&(ptr), (size_t)(bytes), tb, err);         \
    if (rc_ != VITX_OK) return rc_;                                        \
  } while (0)

static int64_t find_param(const vitx_engine* e, const std::string& name) {
  for (auto& p : e->table)
    if (p.name == name) return p.aoff;
  return -1;
}

static int init_dense(vitx_engine* e, Dense& w, const std::string& kname, const std::string& bname, int in, int out, std::string& err,
                      bool operand_copies = true) {
  w.in = in;
  w.out = out;
  w.in_k = (int)round_up(in, 64);
  w.out_k = (int)round_up(out, 64);
  w.w = find_param(e, kname);
  w.b = bname.empty() ? -1 : find_param(e, bname);
  if (w.w < 0) { err = "missing parameter " + kname; return VITX_ERR_INVALID; }
  if (e->bf16 && operand_copies) {
    DALLOC(w.wt, (size_t)round_up(out, 256) * w.in_k * 2, false);
    DALLOC(w.wn, (size_t)round_up(in, 256) * w.out_k * 2, false);
  }
  return VITX_OK;
}

static inline const float* dense_w(const vitx_engine* e, const Dense& w) { return w.ext_w ? w.ext_w : e->params + w.w; }
static inline const float* dense_b(const vitx_engine* e, const Dense& w) { return w.ext_b ? w.ext_b : (w.b >= 0 ? e->params + w.b : nullptr); }
static inline float* dense_gw(const vitx_engine* e, const Dense& w) { return w.ext_gw ? w.ext_gw : e->grads + w.w; }
static inline float* dense_gb(const vitx_engine* e, const Dense& w) { return w.ext_gb ? w.ext_gb : (w.b >= 0 ? e->grads + w.b : nullptr); }

// device table of the batched operand refresh: arena offsets and operand buffers of every Dense kernel.  Built at creation (not inside a step, which
// may be under stream capture) and RE-WRITTEN whenever the parameter arena moves (vitx_bind_arenas hands the library a caller-owned arena: round 6 --
// until then the table kept pointing into the library's own arena, so after a bind the bf16 operand copies were refreshed from weights nobody
// updated any more; rank 0 of a data-parallel run never noticed, every other rank computed with its pre-broadcast kernels.  Found the first time two
// ranks met on the torch exchange: tests/test_gpu_dp.py::test_engine_dp_two_ranks_on_one_gpu_over_gloo_equals_one_rank)
static void fill_convert_table(vitx_engine* e, std::vector<ConvertDesc>& t, int& blocks) {
  t.clear();
  blocks = 0;
  auto add = [&](const Dense& w) {
    if (w.w < 0 || !w.wt) return;
    const int tx = (int)ceil_div(w.out, 64), ty = (int)ceil_div(w.in, 64);
    t.push_back({e->params + w.w, w.wn, w.wt, w.out_k, w.in_k, w.in, w.out, tx, blocks});
    blocks += tx * ty;
  };
  // CaiT patch stage: to_q and to_kv read the same tokens (cait.py:109-116 with context = None) and run as ONE Dense on the concatenated operand
  // copies [to_q | to_kv]; parameters and gradients stay two tensors
  auto add_cat = [&](const BlockParams& b) {
    if (!b.qkvcat.wt) return;
    const Dense& c = b.qkvcat;
    int col = 0;
    for (const Dense* w : {&b.q, &b.kv}) {
      const int tx = (int)ceil_div(w->out, 64), ty = (int)ceil_div(w->in, 64);
      t.push_back({e->params + w->w, c.wn + col, c.wt + (int64_t)col * c.in_k, c.out_k, c.in_k, w->in, w->out, tx, blocks});
      blocks += tx * ty;
      col += w->out;
    }
  };
  add(e->patch);
  add(e->head);
  for (auto& st : e->stages)
    for (auto& b : st.bp) { add(b.qkv); add(b.q); add(b.kv); add_cat(b); add(b.out); add(b.fc1); add(b.fc2); }
}
static void build_convert_table(vitx_engine* e) {
  if (!e->bf16 || e->conv_descs) return;
  std::vector<ConvertDesc> t;
  int blocks = 0;
  fill_convert_table(e, t, blocks);
  if (!t.empty() && hipMalloc(&e->conv_descs, t.size() * sizeof(ConvertDesc)) == hipSuccess) {
    e->allocs.push_back(e->conv_descs);
    (void)hipMemcpy(e->conv_descs, t.data(), t.size() * sizeof(ConvertDesc), hipMemcpyHostToDevice);
    e->conv_n = (int)t.size();
    e->conv_blocks = blocks;
  }
}
// the parameter arena has moved (vitx_bind_arenas): the same table over the new base pointer
void engine_params_moved(vitx_engine* e) {
  e->params_dirty = true;
  if (!e->bf16 || !e->conv_descs) return;
  std::vector<ConvertDesc> t;
  int blocks = 0;
  fill_convert_table(e, t, blocks);
  if ((int)t.size() != e->conv_n) return;   // (cannot happen: the table is a function of the model)
  (void)hipStreamSynchronize(e->stream);   // a refresh in flight still reads the old table
  (void)hipMemcpy(e->conv_descs, t.data(), t.size() * sizeof(ConvertDesc), hipMemcpyHostToDevice);
}

void engine_refresh_weights(vitx_engine* e) {
  if (!e->bf16) { e->params_dirty = false; return; }
  Prof pr(e, "convert_weights", 0, (double)e->n_params * 8);
  if (e->conv_descs) launch_convert_weights_batched((const ConvertDesc*)e->conv_descs, e->conv_n, e->conv_blocks, e->stream);
  else {   // no table (allocation failed): one launch per matrix
    auto conv = [&](const Dense& w) {
      if (w.w >= 0 && w.wt) launch_convert_weight(e->params + w.w, w.in, w.out, w.wn, w.out_k, w.wt, w.in_k, e->stream);
    };
    conv(e->patch);
    conv(e->head);
    for (auto& st : e->stages)
      for (auto& b : st.bp) {
        conv(b.qkv); conv(b.q); conv(b.kv); conv(b.out); conv(b.fc1); conv(b.fc2);
        if (b.qkvcat.wt) {   // [to_q | to_kv] into the concatenated copies
          const Dense& c = b.qkvcat;
          launch_convert_weight(e->params + b.q.w, b.q.in, b.q.out, c.wn, c.out_k, c.wt, c.in_k, e->stream);
          launch_convert_weight(e->params + b.kv.w, b.kv.in, b.kv.out, c.wn + b.q.out, c.out_k, c.wt + (int64_t)b.q.out * c.in_k, c.in_k, e->stream);
        }
      }
  }
  e->params_dirty = false;
}

// profiler class of an all-generic GEMM launch = the kernel launch_gemm_generic will pick
static const char* f32_gemm_class(const GenericGemmArgs& g, int ta, int tb, int to) {
  if (g.x3 && gemm_bf16x3_supported(g, ta, tb, to)) return "gemm_bf16x3_mfma";
  return gemm_f32_mfma_supported(g, ta, tb, to) ? "gemm_f32_mfma" : "gemm_generic_fma";
}


// ------------------------------------------------------------------------------------------------
// side stream: weight gradients beside the input-gradient chain
// ------------------------------------------------------------------------------------------------
// A weight gradient dW = X^T dY has no consumer until the optimizer / the gradient exchange, while everything else in the backward pass is one
// dependent chain (dgrad GEMM -> LayerNorm VJP -> dgrad GEMM ...).  On ONE stream the step is the SUM of its MFMA-bound kernels (GEMMs) and its
// HBM-bound ones (LayerNorm VJP, attention VJP, reductions): the matrix pipes idle under the latter, the memory system under the former, and the
// persistent GEMMs leave CUs idle in their last partial round of tiles.  The weight-gradient GEMMs and their slice reductions therefore go to a
// second stream: forked from the main stream by an event at the point their operands exist, joined back at the end of the backward pass.
//   * operands the chain rewrites every block (d hpre, d qkv, the bf16 residual gradient, the branch gradient) live in small rings (SideRing):
//     the chain writes the NEXT slot, and waits (event) for the side-stream reader of that slot only when it comes round again;
//   * saved activations (y1, y2, act, o, ctx) are not rewritten before the next forward, which is behind the join;
//   * the split-K partial workspace is used by weight gradients only; launches on either stream are ordered by the fork / join events;
//   * gradient-ready reports (data parallel) of a block are delivered one block late, behind a wait for the side stream's share of the range.
// Results are bit-identical to the one-stream order: the same kernels on the same operands, only their interleaving changes.
static hipEvent_t side_event(vitx_engine* e) {
  if (e->side_events.empty()) {
    e->side_events.resize(64);
    for (auto& ev : e->side_events) (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  }
  hipEvent_t ev = e->side_events[e->side_ev_next];
  e->side_ev_next = (e->side_ev_next + 1) % e->side_events.size();
  return ev;
}
// stream a weight gradient is launched on; forks the side stream behind everything queued on the main stream so far
// (`pre`: an event recorded earlier on the main stream, at the point the operands of this weight gradient were complete -- side_prefork)
static hipStream_t side_fork(vitx_engine* e, hipEvent_t pre = nullptr) {
  if (!e->side_live) return e->stream;
  hipEvent_t ev = pre;
  if (!ev) {
    ev = side_event(e);
    (void)hipEventRecord(ev, e->stream);
  }
  (void)hipStreamWaitEvent(e->side, ev, 0);
  e->side_dirty = true;
  return e->side;
}
// The engine queues a weight gradient BEHIND the input-gradient GEMMs that share its operands (they want the operand while the memory-side cache
// still holds it); forked at its own place in the queue it would also wait for those GEMMs.  The fork point is therefore recorded where the
// operands are complete, and handed to dense_wgrad later.  nullptr when the side stream is not in use.
static hipEvent_t side_prefork(vitx_engine* e) {
  if (!e->side_live) return nullptr;
  hipEvent_t ev = side_event(e);
  (void)hipEventRecord(ev, e->stream);
  return ev;
}
// the current slot of `r` is being read by work just queued on the side stream
static void side_note_read(vitx_engine* e, SideRing& r) {
  if (!e->side_live || r.n == 0) return;
  (void)hipEventRecord(r.rd[r.cur], e->side);
  r.pend[r.cur] = true;
}
// next slot of `r` for a producer on the main stream (the pointer the engine uses for this buffer from here on)
template <typename P>
static void side_rotate(vitx_engine* e, SideRing& r, P*& ptr) {
  if (!e->side_live || r.n == 0) return;
  r.cur = (r.cur + 1) % r.n;
  if (r.pend[r.cur]) { (void)hipStreamWaitEvent(e->stream, r.rd[r.cur], 0); r.pend[r.cur] = false; }
  ptr = (P*)r.slot[r.cur];
}
// an arena range is final on the compute stream: the library's own bucketed exchange (comm.hip) and / or the caller's callback
static void notify_ready(vitx_engine* e, int64_t off, int64_t cnt) {
  if (e->cm.overlap) comm_on_ready(e, off, cnt);
  if (e->grad_cb) e->grad_cb(e->grad_cb_user, off, cnt);
}
static void side_flush_ready(vitx_engine* e, size_t keep) {
  while (e->side_ready.size() > keep) {
    PendingReady pr = e->side_ready.front();
    e->side_ready.erase(e->side_ready.begin());
    // a caller-run exchange (callback) orders itself against the compute stream, so the compute stream waits for the side stream's share of the range;
    // the library's own exchange lets its COMMUNICATION stream wait instead (round 6): the input-gradient chain is not held up by the weight gradients
    // of the block it reports (forced DP on one GPU: the waits were most of what the exchange cost there)
    static const int chain_env = [] { const char* v = vitx_env("VITX_COMM_WAIT_ON_CHAIN"); return v ? atoi(v) : 0; }();   // 1: the round-5 rule (A/B)
    if (e->grad_cb || !e->cm.overlap || chain_env) (void)hipStreamWaitEvent(e->stream, pr.ev, 0);
    else comm_wait_event(e, pr.ev);
    notify_ready(e, pr.off, pr.cnt);
  }
}
// gradient-ready report of an arena range whose weight gradients may still be queued on the side stream
static void report_ready(vitx_engine* e, int64_t off, int64_t cnt) {
  if (!e->grad_cb && !e->cm.overlap) return;
  if (!e->side_live) { notify_ready(e, off, cnt); return; }
  hipEvent_t ev = side_event(e);
  (void)hipEventRecord(ev, e->side);
  e->side_ready.push_back({off, cnt, ev});
  // callback: everything but the newest (its side-stream work was queued a block ago: the compute stream's wait is then free);
  // native exchange: at once -- the wait goes to the communication stream, and the bucket leaves a block earlier
  side_flush_ready(e, (e->grad_cb || !e->cm.overlap) ? 1 : 0);
}
static void side_join(vitx_engine* e) {
  if (e->side_dirty) {
    hipEvent_t ev = side_event(e);
    (void)hipEventRecord(ev, e->side);
    (void)hipStreamWaitEvent(e->stream, ev, 0);
    e->side_dirty = false;
  }
  for (SideRing* r : {&e->rg_dh, &e->rg_glp, &e->rg_dqkv, &e->rg_dbr, &e->rg_lnp, &e->rg_cs})
    for (int i = 0; i < SIDE_RING_MAX; ++i) r->pend[i] = false;
  side_flush_ready(e, 0);
}
// scope of a backward pass: decides whether this pass uses the side stream, joins on every way out
struct SideScope {
  vitx_engine* e;
  SideScope(vitx_engine* e_, bool allowed) : e(e_) {
    bool on = allowed && e->side_mode && e->side && e->bf16 && !e->force_generic_gemm && !e->wgrad_via_transpose && !e->profiling;
    if (on) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(e->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) on = false;
    }
    e->side_live = on;
  }
  ~SideScope() {
    if (e->side_live) side_join(e);
    e->side_live = false;
  }
};

// the small-reduction stream: behind everything queued on the main stream so far ...
static void side2_begin(vitx_engine* e) {
  hipEvent_t ev = side_event(e);
  (void)hipEventRecord(ev, e->stream);
  (void)hipStreamWaitEvent(e->side2, ev, 0);
}
// ... and, once its launches are queued: the ring slot they read is released by an event on that stream, and the weight-gradient stream is
// ordered behind them, so that the join and the gradient-ready reports (events on that stream) cover them too
static void side2_end(vitx_engine* e, SideRing& r) {
  (void)hipEventRecord(r.rd[r.cur], e->side2);
  r.pend[r.cur] = true;
  (void)hipStreamWaitEvent(e->side, r.rd[r.cur], 0);
  e->side_dirty = true;
}
// LayerNorm VJP of a block: with the side stream live, the two small launches that reduce its per-block partials into dgamma / dbeta (/ a bias
// gradient) leave the input-gradient chain -- 5 us each, 50 per ViT-B/16 step, nothing downstream reads their results.  They go to a stream of
// their own: on the weight-gradient stream they queued behind a block's GEMMs and piled up in the tail after the chain had finished (CaiT cfg5
// +0.2 .. 0.6 ms per step, r4x); here they run as soon as the VJP kernel is done, in whatever gaps the chip has.  The weight-gradient stream is
// then ordered behind them, so that the join and the gradient-ready reports (events on that stream) cover them too.
// (round 5) `nb` != null: the branch that consumes this VJP's result has a LayerScale (CaiT) -- its VJP rides on this pass (layernorm_bwd_kernel, FUSE):
// e->d_br receives g * scale, the scale gradient and that branch's bias gradient come out of two more partial rows
struct NextBranch { const void* f; const float* scale; float* dscale; float* dbias; };
static void block_layernorm_bwd(vitx_engine* e, const void* dy, int T, int d, const float* x, const float* mean, const float* rstd, const float* gamma,
                                const float* g_in, float* g_out, void* g_lp, float* dgamma, float* dbeta, float* gsum, int rows,
                                const NextBranch* nb = nullptr) {
  if (nb) {
    const bool side = e->side_live && e->side2 && e->rg_lnp.n != 0 && rows >= e->side2_min_rows;
    if (side) side_rotate(e, e->rg_lnp, e->ln_part);
    float* part = side ? e->ln_part : e->red_ws;
    int parts = 0;
    launch_layernorm_bwd_scale((const bf16_t*)dy, d, x, d, mean, rstd, gamma, g_in, d, g_out, d, (bf16_t*)e->d_br, d, (const bf16_t*)nb->f, d, nb->scale,
                               part, rows, d, e->stream, &parts);
    if (side) side2_begin(e);
    launch_layernorm_bwd_scale_reduce(part, parts, d, dgamma, dbeta, nb->dscale, nb->dbias, nb->scale, side ? e->side2 : e->stream);
    if (side) side2_end(e, e->rg_lnp);
    return;
  }
  if (!e->side_live || !e->side2 || e->rg_lnp.n == 0 || rows < e->side2_min_rows) {   // short VJPs (CaiT's class-attention rows): three stream operations cost more than they hide
    launch_layernorm_bwd(dy, T, d, x, d, mean, rstd, gamma, g_in, d, g_out, d, g_lp, d, e->red_ws, dgamma, dbeta, gsum, rows, d, e->stream);
    return;
  }
  side_rotate(e, e->rg_lnp, e->ln_part);
  int parts = 0;
  launch_layernorm_bwd(dy, T, d, x, d, mean, rstd, gamma, g_in, d, g_out, d, g_lp, d, e->ln_part, dgamma, dbeta, gsum, rows, d, e->stream, &parts);
  side2_begin(e);
  launch_layernorm_bwd_reduce(e->ln_part, parts, d, dgamma, dbeta, (gsum && g_in) ? gsum : nullptr, e->side2);
  side2_end(e, e->rg_lnp);
}

// ------------------------------------------------------------------------------------------------
// Dense layer = x @ kernel[in,out] + bias (Keras nn.Dense; vit.py:39,42,59,63,143,156) and its VJPs


// Algorithmic HBM bytes of one fused Dense launch: every operand read once and every output written once AT ITS STORAGE WIDTH, the operands of the
// fused epilogue included (the fp32 residual read + fp32 result of vit.py:101-102, the two bf16 outputs of the fc1 epilogue, the stored gelu' the
// fc2 input gradient multiplies by).  Until round 4 this was A + W + one bf16 output for every launch, which under-counted the fused launches by
// 1.4x (423 against 306 MB per launch of the family at ViT-B/16, batch 256) -- and made the PMC traffic look 1.8x "wasted" where it is 1.3x.
static double dense_launch_bytes(const vitx_engine* e, int mode, const EpiParams& ep, double rows, double k, double n) {
  const double esz = e->esz;
  double b = rows * k * esz + k * n * esz;                      // A, W
  switch (mode) {
    case EPI_BIAS_GELU: b += 2.0 * rows * n * esz; break;        // act (or h) and gelu'(h)
    case EPI_BIAS_RESID: b += 2.0 * rows * n * 4.0 + ((ep.scale && ep.out2) ? rows * n * esz : 0.0); break;   // fp32 residual in, fp32 out (+ LayerScale's f)
    case EPI_GELU_BWD: b += 2.0 * rows * n * esz; break;         // stored gelu' (or h) in, d h out
    case EPI_STORE_F32: case EPI_PATCH: b += rows * n * 4.0; break;
    default: b += rows * n * esz; break;
  }
  return b;
}

// ------------------------------------------------------------------------------------------------
static void dense_fwd(vitx_engine* e, const void* X, int64_t ldx, int rows, const Dense& w, int mode, EpiParams ep) {
  ep.M = rows;
  ep.N = w.out;
  if (mode != EPI_GELU_BWD) ep.bias = dense_b(e, w);
  const double flops = 2.0 * rows * (double)w.out * w.in;
  const double bytes = dense_launch_bytes(e, mode, ep, rows, w.in, w.out);
  if (e->bf16 && !e->force_generic_gemm) {
    Bf16GemmArgs g;
    g.A = (const bf16_t*)X; g.lda = ldx;
    g.B = w.wt; g.ldb = w.in_k;
    g.M = rows; g.N = w.out; g.K = w.in_k; g.kernel = e->gemm_kernel; g.tail = e->gemm_tail;
    g.reverse_m = (e->reverse_mask & 1) && (int64_t)rows * w.in_k * 2 > e->reverse_min_bytes;   // A larger than the memory-side cache, just written
    g.stagger = (mode == EPI_BIAS_GELU || mode == EPI_BIAS_RESID || mode == EPI_PATCH) ? e->gemm_stagger : 0;
    g.shared_gpu = comm_busy(e);
    ep.zero_pad = 1;
    finalize_epi(ep);
    char shape[48];
    if (e->profiling) snprintf(shape, sizeof shape, "shape nt e%d %dx%dx%d", mode, g.M, g.N, g.K);   // per-shape row of bench.py's gemm_shapes table
    Prof pr(e, "gemm_bf16_mfma", flops, bytes, e->profiling ? shape : nullptr);
    launch_gemm_bf16(g, ep, mode, e->stream);
  } else {
    GenericGemmArgs g;
    g.A = X; g.B = dense_w(e, w);
    g.M = rows; g.N = w.out; g.K = w.in;
    g.sam = ldx; g.sak = 1; g.sbk = w.out; g.sbn = 1; g.x3 = e->x3;
    ep.zero_pad = 1;
    finalize_epi(ep);
    Prof pr(e, f32_gemm_class(g, e->bf16, 0, e->bf16), flops, bytes);   // the class of the kernel that runs
    launch_gemm_generic(g, ep, mode, e->bf16, 0, e->bf16, e->stream);
  }
}

// dX[rows,in] = dY[rows,out] @ W^T
static void dense_dgrad(vitx_engine* e, const void* dY, int64_t ldy, int rows, const Dense& w, int mode, EpiParams ep) {
  ep.M = rows;
  ep.N = w.in;
  ep.bias = nullptr;
  const double flops = 2.0 * rows * (double)w.out * w.in;
  const double bytes = dense_launch_bytes(e, mode, ep, rows, w.out, w.in);
  if (e->bf16 && !e->force_generic_gemm) {
    Bf16GemmArgs g;
    g.A = (const bf16_t*)dY; g.lda = ldy;
    g.B = w.wn; g.ldb = w.out_k;
    g.M = rows; g.N = w.in; g.K = w.out_k; g.kernel = e->gemm_kernel; g.tail = e->gemm_tail;
    g.reverse_m = (e->reverse_mask & 2) && (int64_t)rows * w.out_k * 2 > e->reverse_min_bytes;
    g.stagger = (mode == EPI_GELU_BWD) ? e->gemm_stagger : 0;
    g.shared_gpu = comm_busy(e);   // a bucket's collective may be running beside this launch (native exchange, comm.hip)
    ep.zero_pad = 1;
    finalize_epi(ep);
    char shape[48];
    if (e->profiling) snprintf(shape, sizeof shape, "shape nt e%d %dx%dx%d", mode, g.M, g.N, g.K);
    Prof pr(e, "gemm_bf16_mfma", flops, bytes, e->profiling ? shape : nullptr);
    launch_gemm_bf16(g, ep, mode, e->stream);
  } else {
    GenericGemmArgs g;
    g.A = dY; g.B = dense_w(e, w);
    g.M = rows; g.N = w.in; g.K = w.out;
    g.sam = ldy; g.sak = 1; g.sbk = 1; g.sbn = w.out; g.x3 = e->x3;
    ep.zero_pad = 1;
    finalize_epi(ep);
    Prof pr(e, f32_gemm_class(g, e->bf16, 0, e->bf16), flops, bytes);   // the class of the kernel that runs
    launch_gemm_generic(g, ep, mode, e->bf16, 0, e->bf16, e->stream);
  }
}

static void dense_wgrad_generic(vitx_engine* e, const void* X, int64_t ldx, const void* dY, int64_t ldy, int rows, const Dense& w) {
  float* dW = dense_gw(e, w);
  const double flops = 2.0 * rows * (double)w.out * w.in;
  const double bytes = (double)rows * w.in * e->esz + (double)rows * w.out * e->esz + (double)w.in * w.out * 4;
  {
    GenericGemmArgs g;
    g.A = X; g.B = dY;
    g.M = w.in; g.N = w.out; g.K = rows;
    g.sam = 1; g.sak = ldx; g.sbk = ldy; g.sbn = 1; g.x3 = e->x3;
    EpiParams ep;
    ep.out = dW; ep.ldo = w.out; ep.M = w.in; ep.N = w.out;
    // BF16X3 mode: a weight gradient is a handful of 128 x 128 tiles over tens of thousands of token rows -- split the rows into slices (the batch
    // index of the kernel) so that tiles x slices fills the chip twice over, fp32 partials + the reduction pass of the bf16 mode
    int slices = 1;
    if (e->x3 && e->partial_ws && gemm_bf16x3_supported(g, 0, 0, 0)) {
      const int64_t tiles = ceil_div(w.in, 128) * ceil_div(w.out, 128);
      int64_t want = std::min<int64_t>({std::max<int64_t>(1, 1024 / tiles), std::max<int64_t>(1, rows / 256), e->partial_elems / ((int64_t)w.in * w.out), 32});   // (<= 32: the two-level reduction)
      if (want > 1) {
        const int ks = (int)round_up(ceil_div(rows, want), 32);
        slices = (int)ceil_div(rows, ks);
        if (slices > 1) {
          g.K = ks; g.nb = slices; g.sAb = (int64_t)ks * ldx; g.sBb = (int64_t)ks * ldy; g.k_last = rows - (slices - 1) * ks;
          ep.out = e->partial_ws; ep.out_batch_stride = (int64_t)w.in * w.out;
        }
      }
    }
    finalize_epi(ep);
    {
      Prof pr(e, f32_gemm_class(g, e->bf16, e->bf16, 0), flops, bytes);   // the class of the kernel that runs
      launch_gemm_generic(g, ep, EPI_STORE_F32, e->bf16, e->bf16, 0, e->stream);
    }
    if (slices > 1) {
      Prof pr(e, "reduce_partials", 0, (double)(slices + 1) * w.in * w.out * 4);
      launch_reduce_partials(e->partial_ws, slices, (int64_t)w.in * w.out, (int64_t)w.in * w.out, dW, 1.0f, e->stream);
    }
  }
}

// dW[in,out] = X^T[in,rows] @ dY[rows,out]   (reduction over every token row of the batch)
static void dense_wgrad(vitx_engine* e, const void* X, int64_t ldx, const void* dY, int64_t ldy, int rows, const Dense& w, hipEvent_t pre = nullptr) {
  float* dW = dense_gw(e, w);
  const double flops = 2.0 * rows * (double)w.out * w.in;
  const double bytes = (double)rows * w.in * e->esz + (double)rows * w.out * e->esz + (double)w.in * w.out * 4;
  if (e->bf16 && !e->force_generic_gemm) {
    const hipStream_t ws = side_fork(e, pre);   // the side stream inside a backward pass that uses it, else the main stream
    const int kext = (int)round_up(rows, 64);   // rows >= `rows` of both operands are zero (row-padding invariant)
    Bf16GemmArgs g;
    g.M = w.in; g.N = w.out; g.K = kext; g.kernel = e->gemm_kernel;
    int tm, tn;
    if (e->wgrad_via_transpose) {
      Prof pr(e, "transpose_bf16", 0, 2.0 * ((double)kext * w.in + (double)kext * w.out) * 2);
      launch_transpose_bf16((const bf16_t*)X, ldx, kext, w.in, e->xt, kext, ws);
      launch_transpose_bf16((const bf16_t*)dY, ldy, kext, w.out, e->dyt, kext, ws);
      g.A = e->xt; g.lda = kext;
      g.B = e->dyt; g.ldb = kext;
      tm = gemm_bf16_tile_m(g.kernel, g.M, g.N); tn = gemm_bf16_tile_n(g.kernel, g.M, g.N);
    } else {
      g.A = (const bf16_t*)X; g.lda = ldx;
      g.B = (const bf16_t*)dY; g.ldb = ldy;
      // few token rows (small batches): the K extent cannot be split, so 256x256 tiles leave most CUs idle and each workgroup's
      // time is its prologue plus a 256-KiB partial store; 128x128 tiles give four times the workgroups and a quarter of that store
      if (g.kernel == 0 && ceil_div(w.in, 256) * ceil_div(w.out, 256) * std::max<int64_t>(1, ceil_div(kext / 64, 4)) <= 128) g.kernel = 1;
      tm = tn = gemm_bf16_tn_tile(g.kernel, g.M, g.N);
    }
    const int64_t tiles = ceil_div(w.in, tm) * ceil_div(w.out, tn);
    const int nk = kext / 64;
    // one wave of workgroups (these kernels run 1 WG/CU): split-K so that tiles*split ~ 256, fewer slices = less partial traffic
    // (finer slices -- 384 / 512 workgroups, so that a low-priority side-stream workgroup holds its CU for less long -- measured +2.5 / +1.6 ms per step:
    //  profiles/r4/ab_weight_gradient_slices_r4o_not_kept.log; coarser ones -- 192 / 128 workgroups, less partial traffic -- +0.3 / +0.8 ms:
    //  ab_weight_gradient_coarser_slices_r4pj_not_kept.log)
    // (round 5) with few token rows -- 16 k rows: CaiT / DeepViT at 256 x 64 tokens -- a full wave of workgroups means 16 K-tiles per slice, a partial
    //  buffer as large as the operands, and a launch that takes every CU from the input-gradient GEMMs it runs beside; HALF a wave leaves those
    //  their CUs and halves the partial traffic: CaiT cfg5 38.4 -> 36.7 ms, README config -2 %, DeepViT cfg4 neutral; at 50 k rows (ViT-B / L) it
    //  costs 1.0 / 2.0 ms (profiles/r5/sweep_weight_gradient_workgroups_r5t.log).  VITX_WGRAD_WGS=n overrides.
    static const int wg_env = [] { const char* v = vitx_env("VITX_WGRAD_WGS"); return v ? std::max(1, atoi(v)) : 0; }();
    const int wg_target = wg_env ? wg_env : (rows <= 24576 ? 128 : 256);
    int split = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(nk, 4), std::max<int64_t>(1, wg_target / tiles)));
    while (split > 1 && (int64_t)split * w.in * w.out > e->partial_elems) --split;
    if (!e->wgrad_via_transpose) {
      // the weight-gradient kernel addresses a K slice of each operand through a buffer resource (31-bit byte offsets): a slice must stay below
      // 2 GiB -- more slices if it would not (never at the shapes of the models here: 50k rows x 4096 features is 0.4 GB)
      const int64_t max_rows = (((1LL << 31) - (1 << 20)) / (2 * std::max<int64_t>(ldx, ldy))) / 64 * 64;
      split = (int)std::max<int64_t>(split, ceil_div(kext, std::max<int64_t>(64, max_rows)));
      //  the raised slice count must still fit the partial buffer: if it does not, this one launch takes the strided fp32-FMA kernel
      // (64-bit addressing, no partial buffer; slow and correct -- beyond 262k token rows x 4096 features, no model of the reference gets there)
      if (split > 1 && (int64_t)split * w.in * w.out > e->partial_elems) {
        dense_wgrad_generic(e, X, ldx, dY, ldy, rows, w);
        return;
      }
    }
    g.split_k = split;
    const int slices = gemm_bf16_num_slices(kext, split);
    EpiParams ep;
    // a single K slice (small batches: few token rows) is the gradient itself: written in place, no partial buffer, no reduction pass
    ep.out = slices == 1 ? dW : e->partial_ws; ep.ldo = w.out; ep.partial_stride = (int64_t)w.in * w.out;
    ep.M = w.in; ep.N = w.out;
    finalize_epi(ep);
    {
      char shape[48];
      if (e->profiling) snprintf(shape, sizeof shape, "shape tn s%d %dx%dx%d", slices, g.M, g.N, g.K);   // dW[in, out] over K token rows
      Prof pr(e, e->wgrad_via_transpose ? "gemm_bf16_mfma" : "gemm_bf16_mfma_tn", flops, bytes, e->profiling ? shape : nullptr, ws);
      if (e->wgrad_via_transpose) launch_gemm_bf16(g, ep, EPI_PARTIAL, ws);
      else launch_gemm_bf16_tn(g, ep, ws);
    }
    if (slices > 1) {
      Prof pr(e, "reduce_partials", 0, (double)(slices + 1) * w.in * w.out * 4, nullptr, ws);
      launch_reduce_partials(e->partial_ws, slices, (int64_t)w.in * w.out, (int64_t)w.in * w.out, dW, 1.0f, ws);
    }
  } else {
    dense_wgrad_generic(e, X, ldx, dY, ldy, rows, w);
  }
}

static void bias_grad(vitx_engine* e, const void* dY, int is_bf16, int64_t ld, int rows, const Dense& w) {
  float* gb = dense_gb(e, w);
  if (!gb) return;
  Prof pr(e, "colsum", 0, (double)rows * w.out * (is_bf16 ? 2 : 4));
  launch_colsum(dY, is_bf16, ld, rows, w.out, e->red_ws, gb, e->stream);
}

// ---- Dense layers of a wrapper object (MAE enc_to_dec / to_pixels, SimMIM to_pixels: mae.py:41,45, simmim.py:84) on this engine
int engine_ext_dense_init(vitx_engine* e, Dense& w, int in, int out, const float* W, const float* bias, float* gW, float* gbias, std::string& err) {
  w.in = in; w.out = out;
  w.in_k = (int)round_up(in, 64); w.out_k = (int)round_up(out, 64);
  w.w = -1; w.b = -1;
  w.ext_w = W; w.ext_b = bias; w.ext_gw = gW; w.ext_gb = gbias;
  if (e->bf16) {
    DALLOC(w.wt, (size_t)round_up(out, 256) * w.in_k * 2, false);
    DALLOC(w.wn, (size_t)round_up(in, 256) * w.out_k * 2, false);
  }
  return VITX_OK;
}
void engine_ext_dense_refresh(vitx_engine* e, const Dense& w) {
  if (e->bf16 && w.wt) launch_convert_weight(w.ext_w, w.in, w.out, w.wn, w.out_k, w.wt, w.in_k, e->stream);
}
void engine_ext_dense_fwd(vitx_engine* e, const void* X, int64_t ldx, int rows, const Dense& w, float* y) {
  EpiParams ep; ep.out = y; ep.ldo = w.out;
  dense_fwd(e, X, ldx, rows, w, EPI_STORE_F32, ep);
}
void engine_ext_dense_bwd(vitx_engine* e, const void* X, int64_t ldx, const void* dY, int64_t ldy, const float* dY_f32, int rows, const Dense& w,
                          float* dx) {
  if (dx) {
    EpiParams ep; ep.out = dx; ep.ldo = w.in;
    dense_dgrad(e, dY, ldy, rows, w, EPI_STORE_F32, ep);
  }
  dense_wgrad(e, X, ldx, dY, ldy, rows, w);
  (void)dY_f32;   // the bias gradient (column sums of the fp32 dY) is the caller's: it sizes its own reduction workspace
}

// ------------------------------------------------------------------------------------------------
// attention
// ------------------------------------------------------------------------------------------------
struct AttnView {   // per-head-addressable views into [b, n, (h d)]-style tensors (vit.py:74,82 as addressing)
  const void *q = nullptr, *k = nullptr, *v = nullptr;
  int64_t ldq = 0, ldk = 0, ldv = 0, qb = 0, kb = 0, vb = 0;   // row strides / per-image strides (elements)
  void* o = nullptr; int64_t ldo = 0, ob = 0;
  int nq = 0, nk = 0;
};

static int ensure_scores(vitx_engine* e, int count, int64_t elems, std::string& err) {
  if (elems > e->sc_elems) {
    // (re)allocate all score workspaces at the larger size
    for (int i = 0; i < 4; ++i) e->sc[i] = nullptr;
    e->sc_elems = elems;
  }
  for (int i = 0; i < count; ++i)
    if (!e->sc[i]) DALLOC(e->sc[i], (size_t)e->sc_elems * 4, false);
  return VITX_OK;
}

struct BgemmCall {   // one batched small product, as bgemm() takes it
  const void* A; int ta; int64_t sam, sak, sAb, sAh; const void* B; int tb; int64_t sbk, sbn, sBb, sBh; int M, N, K, nb, nh, mode, to; void* out;
  int64_t ldo, ob, oh; float alpha;
};
static void bgemm_fill(const BgemmCall& c, GenericGemmArgs& g, EpiParams& ep) {
  g.A = c.A; g.B = c.B; g.M = c.M; g.N = c.N; g.K = c.K;
  g.sam = c.sam; g.sak = c.sak; g.sbk = c.sbk; g.sbn = c.sbn;
  g.nb = c.nb; g.nh = c.nh; g.sAb = c.sAb; g.sAh = c.sAh; g.sBb = c.sBb; g.sBh = c.sBh;
  ep.out = c.out; ep.ldo = c.ldo; ep.out_batch_stride = c.ob; ep.out_head_stride = c.oh; ep.alpha = c.alpha;
  ep.M = c.M; ep.N = c.N;
  finalize_epi(ep);
}
static void bgemm(vitx_engine* e, const void* A, int ta, int64_t sam, int64_t sak, int64_t sAb, int64_t sAh, const void* B, int tb,
                  int64_t sbk, int64_t sbn, int64_t sBb, int64_t sBh, int M, int N, int K, int nb, int nh, int mode, int to, void* out,
                  int64_t ldo, int64_t ob, int64_t oh, float alpha);
// two products of the same (image, head) that share an operand: one launch when both fit the MFMA kernel (attn_bgemm_mfma.hip)
static void bgemm_pair(vitx_engine* e, const BgemmCall& c1, const BgemmCall& c2) {
  GenericGemmArgs g1, g2;
  EpiParams ep1, ep2;
  bgemm_fill(c1, g1, ep1);
  bgemm_fill(c2, g2, ep2);
  if (e->bf16 && !e->force_generic_gemm && e->bgemm_pairs && c1.nb == c2.nb && c1.nh == c2.nh && bgemm_mfma_supported(g1, c1.ta, c1.tb, c1.to, c1.mode) &&
      bgemm_mfma_supported(g2, c2.ta, c2.tb, c2.to, c2.mode)) {
    Prof pr(e, "attn_bgemm_mfma", 2.0 * c1.nb * c1.nh * ((double)c1.M * c1.N * c1.K + (double)c2.M * c2.N * c2.K), 0);
    launch_bgemm_mfma_pair(g1, ep1, c1.ta, c1.to, g2, ep2, c2.ta, c2.to, e->stream);
    return;
  }
  for (const BgemmCall* c : {&c1, &c2})
    bgemm(e, c->A, c->ta, c->sam, c->sak, c->sAb, c->sAh, c->B, c->tb, c->sbk, c->sbn, c->sBb, c->sBh, c->M, c->N, c->K, c->nb, c->nh, c->mode, c->to, c->out,
          c->ldo, c->ob, c->oh, c->alpha);
}
static void bgemm(vitx_engine* e, const void* A, int ta, int64_t sam, int64_t sak, int64_t sAb, int64_t sAh, const void* B, int tb,
                  int64_t sbk, int64_t sbn, int64_t sBb, int64_t sBh, int M, int N, int K, int nb, int nh, int mode, int to, void* out,
                  int64_t ldo, int64_t ob, int64_t oh, float alpha) {
  GenericGemmArgs g;
  g.A = A; g.B = B; g.M = M; g.N = N; g.K = K;
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn;
  g.nb = nb; g.nh = nh; g.sAb = sAb; g.sAh = sAh; g.sBb = sBb; g.sBh = sBh;
  g.x3 = e->x3 && e->x3_attn;
  EpiParams ep;
  ep.out = out; ep.ldo = ldo; ep.out_batch_stride = ob; ep.out_head_stride = oh; ep.alpha = alpha;
  ep.M = M; ep.N = N;
  finalize_epi(ep);
  if (e->bf16 && !e->force_generic_gemm && bgemm_mfma_supported(g, ta, tb, to, mode)) {
    Prof pr(e, "attn_bgemm_mfma", 2.0 * M * (double)N * K * nb * nh, 0);
    launch_bgemm_mfma(g, ep, ta, to, e->stream);
    return;
  }
  Prof pr(e, "attn_generic_bgemm", 2.0 * M * (double)N * K * nb * nh, 0);
  launch_gemm_generic(g, ep, mode, ta, tb, to, e->stream);
}

// forward chain into the score workspaces; returns the index of the workspace holding the matrix that multiplies V
static int attn_generic_scores(vitx_engine* e, const BlockParams& bp, const AttnView& a, int b, bool for_bwd, float* const* sc) {
  const int h = e->cfg.heads, dh = e->cfg.dim_head;
  const int T = e->bf16;
  const int64_t ld = round_up(a.nk, 4);
  const int64_t hs = (int64_t)a.nq * ld, bs = (int64_t)h * hs;
  const float scale = 1.0f / std::sqrt((float)dh);
  // dots = q k^T * scale   (vit.py:77, deepvit.py:79, cait.py:121)
  bgemm(e, a.q, T, a.ldq, 1, a.qb, dh, a.k, T, 1, a.ldk, a.kb, dh, a.nq, a.nk, dh, b, h, EPI_STORE_F32, 0, sc[0], ld, bs, hs, scale);
  const int64_t rows = (int64_t)b * h * a.nq;
  const bool chain = !e->unfused_headops && headchain_supported(h, a.nk);
  if (chain && e->cfg.variant == VITX_VARIANT_CAIT) {
    Prof pr(e, "attn_headchain", 0, 0);     // mix -> softmax -> mix in one pass (cait.py:123-125); A1 is only kept for the backward
    launch_cait_chain_fwd(sc[0], e->params + bp.mix_pre, e->params + bp.mix_post, for_bwd ? sc[1] : nullptr, sc[2], b, h, a.nq, a.nk,
                          ld, e->stream);
    return 2;
  }
  if (chain && e->cfg.variant == VITX_VARIANT_DEEPVIT) {
    Prof pr(e, "attn_headchain", 0, 0);     // softmax -> re-attention mix -> LayerNorm over heads (deepvit.py:80-84)
    launch_deepvit_chain_fwd(sc[0], e->params + bp.re_w, e->params + bp.re_g, e->params + bp.re_b, for_bwd ? sc[1] : nullptr, sc[2],
                             for_bwd ? 1 : 0, b, h, a.nq, a.nk, ld, e->cfg.ln_eps, e->stream);
    return 2;
  }
  if (!e->unfused_headops && e->cfg.variant == VITX_VARIANT_DEEPVIT && deepvit_point_fwd_supported(h, a.nk)) {
    Prof pr(e, "attn_headchain", 0, 0);     // 65..128 keys: row statistics + one fused point kernel (deepvit.py:80-84)
    launch_deepvit_point_fwd(sc[0], e->red_ws, e->params + bp.re_w, e->params + bp.re_g, e->params + bp.re_b, sc[1], sc[2],
                             for_bwd ? 1 : 0, b, h, a.nq, a.nk, ld, e->cfg.ln_eps, e->stream);
    return 2;
  }
  if (e->cfg.variant == VITX_VARIANT_CAIT) {
    Prof pr(e, "attn_generic_headops", 0, 0);
    launch_headmix_fwd(sc[0], e->params + bp.mix_pre, sc[1], b, h, a.nq, a.nk, ld, e->stream);    // cait.py:123
    launch_softmax_rows(sc[1], rows, a.nk, ld, e->stream);                                            // cait.py:124
    launch_headmix_fwd(sc[1], e->params + bp.mix_post, sc[2], b, h, a.nq, a.nk, ld, e->stream);   // cait.py:125
    return 2;
  }
  {
    Prof pr(e, "attn_generic_softmax", 0, 0);
    launch_softmax_rows(sc[0], rows, a.nk, ld, e->stream);                                            // vit.py:78
  }
  if (e->cfg.variant == VITX_VARIANT_DEEPVIT) {
    Prof pr(e, "attn_generic_headops", 0, 0);
    launch_headmix_fwd(sc[0], e->params + bp.re_w, sc[1], b, h, a.nq, a.nk, ld, e->stream);       // deepvit.py:83
    launch_headnorm_fwd(sc[1], e->params + bp.re_g, e->params + bp.re_b, sc[2], b, h, a.nq, a.nk, ld, e->cfg.ln_eps,
                        e->stream);                                                                      // deepvit.py:84
    return 2;
  }
  return 0;
}

static inline int64_t score_geom(int b, int nq, int nk) { return ((int64_t)b << 40) | ((int64_t)nq << 20) | (int64_t)nk; }

// keep != null: the score tensors go to that block's own buffers and stay there for its backward
static void attn_generic_fwd(vitx_engine* e, const BlockParams& bp, const AttnView& a, int b, BlockActs* keep) {
  const int h = e->cfg.heads, dh = e->cfg.dim_head, T = e->bf16;
  const int64_t ld = round_up(a.nk, 4);
  const int64_t hs = (int64_t)a.nq * ld, bs = (int64_t)h * hs;
  float* const* sc = keep ? keep->sc_keep : e->sc;
  if (T && e->cfg.variant == VITX_VARIANT_DEEPVIT && e->deepvit_fused && !e->unfused_headops && !e->force_generic_gemm &&
      deepvit_attn_fused_supported(h, dh, a.nq, a.nk)) {
    // deepvit.py:79-88 in one kernel; the three score tensors only leave the chip when this block's backward will read them
    const double pts = (double)b * h * a.nq * a.nk;
    // (round 5) the normalised scores are kept as bf16 when the one-kernel backward will be their reader (its dV product rounds them anyway)
    const bool lp = e->deepvit_fused_bwd && e->score_bf16;
    Prof pr(e, "attn_deepvit_fused_fwd", 4.0 * pts * dh + 2.0 * pts * h, ((double)b * a.nq * 3 * h * dh + (double)b * a.nq * h * dh) * 2 + (keep ? 8.0 * pts : 0.0));
    launch_deepvit_attn_fwd((const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, a.ldq, a.ldk, a.ldv, a.qb, a.kb, a.vb, (bf16_t*)a.o,
                            a.ldo, a.ob, e->params + bp.re_w, e->params + bp.re_g, e->params + bp.re_b, sc[0], sc[2], keep ? (lp ? 2 : 1) : 0,
                            b, h, a.nq, a.nk, ld, 1.0f / std::sqrt((float)dh), e->cfg.ln_eps, (const bf16_t*)e->zero_page, e->stream);
    if (keep) { keep->sc_geom = score_geom(b, a.nq, a.nk); keep->sc_pi = 2; keep->sc_no_mixed = true; keep->sc_a2_bf16 = lp; }
    return;
  }
  if (T && e->cfg.variant == VITX_VARIANT_CAIT && e->cait_fused && !e->unfused_headops && !e->force_generic_gemm &&
      cait_attn_fused_supported(h, dh, a.nq, a.nk)) {
    // cait.py:121-128 in one kernel (attn_cait_fused.hip); the score tensors are written once, for this block's backward only (the class-attention
    // stage, one query per image, keeps the launch-per-op path)
    const double pts = (double)b * h * a.nq * a.nk;
    Prof pr(e, "attn_cait_fused_fwd", 4.0 * pts * dh + 4.0 * pts * h, ((double)b * a.nq * 3 * h * dh + (double)b * a.nq * h * dh) * 2 + (keep ? 12.0 * pts : 0.0));
    launch_cait_attn_fwd((const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, a.ldq, a.ldk, a.ldv, a.qb, a.kb, a.vb, (bf16_t*)a.o, a.ldo, a.ob,
                         e->params + bp.mix_pre, e->params + bp.mix_post, sc[0], sc[1], sc[2], keep ? (e->score_bf16 ? 2 : 1) : 0, b, h, a.nq, a.nk, ld,
                         1.0f / std::sqrt((float)dh), (const bf16_t*)e->zero_page, e->stream);
    if (keep) { keep->sc_geom = score_geom(b, a.nq, a.nk); keep->sc_pi = 2; keep->sc_no_mixed = false; keep->sc_a2_bf16 = e->score_bf16; }
    return;
  }
  const int pi = attn_generic_scores(e, bp, a, b, keep != nullptr, sc);
  if (keep) { keep->sc_geom = score_geom(b, a.nq, a.nk); keep->sc_pi = pi; keep->sc_no_mixed = false; keep->sc_a2_bf16 = false; }
  // out = attn v   (vit.py:81, deepvit.py:87, cait.py:127)
  bgemm(e, sc[pi], 0, ld, 1, bs, hs, a.v, T, a.ldv, 1, a.vb, dh, a.nq, dh, a.nk, b, h, EPI_STORE, T, a.o, a.ldo, a.ob, dh, 1.0f);
}

struct AttnGrad {   // gradients, same addressing conventions as AttnView
  const void* d_o = nullptr; int64_t ldo = 0, ob = 0;
  void *dq = nullptr, *dk = nullptr, *dv = nullptr;
  int64_t lddq = 0, lddk = 0, lddv = 0, dqb = 0, dkb = 0, dvb = 0;
};

static void attn_generic_bwd(vitx_engine* e, const BlockParams& bp, const AttnView& a, const AttnGrad& gr, int b, BlockActs* keep) {
  const int h = e->cfg.heads, dh = e->cfg.dim_head, T = e->bf16;
  const int64_t ld = round_up(a.nk, 4);
  const int64_t hs = (int64_t)a.nq * ld, bs = (int64_t)h * hs;
  const float scale = 1.0f / std::sqrt((float)dh);
  const int64_t rows = (int64_t)b * h * a.nq;
  // the forward chain: as the block's forward left it, or recomputed (kept tensors absent / describing another geometry)
  const bool kept = keep && keep->sc_keep[0] && keep->sc_geom == score_geom(b, a.nq, a.nk);
  float* const* sc = kept ? keep->sc_keep : e->sc;
  int pi;
  bool a2_lp = kept && keep->sc_a2_bf16;   // sc[pi] holds bf16 [nq][ld2] planes (the one-kernel forwards; DeepViT's is consumed by its own branch below)
  if (kept) {
    pi = keep->sc_pi;
  } else if (T && e->cfg.variant == VITX_VARIANT_CAIT && e->cait_fused && !e->unfused_headops && !e->force_generic_gemm &&
             cait_attn_fused_supported(h, dh, a.nq, a.nk)) {
    // score tensors not kept (budget): recomputed by the kernel that produced them in the forward (same bits as the kept ones), without its A V stage
    Prof pr(e, "attn_cait_fused_fwd", 0, 0);
    launch_cait_attn_fwd((const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, a.ldq, a.ldk, a.ldv, a.qb, a.kb, a.vb, nullptr, 0, 0,
                         e->params + bp.mix_pre, e->params + bp.mix_post, e->sc[0], e->sc[1], e->sc[2], e->score_bf16 ? 2 : 1, b, h, a.nq, a.nk, ld,
                         1.0f / std::sqrt((float)dh), (const bf16_t*)e->zero_page, e->stream);
    pi = 2;
    a2_lp = e->score_bf16;
  } else {
    pi = attn_generic_scores(e, bp, a, b, true, e->sc);
  }
  float* dA = e->sc[3];
  if (T && e->cfg.variant == VITX_VARIANT_DEEPVIT && e->deepvit_fused && e->deepvit_fused_bwd && kept && keep->sc_no_mixed && !e->unfused_headops &&
      !e->force_generic_gemm && deepvit_attn_fused_supported(h, dh, a.nq, a.nk)) {
    // deepvit.py:79-88 backwards in ONE kernel up to d(q) (attn_deepvit_fused.hip): d(attn') = dO v^T, LayerNorm-over-heads VJP, mix VJP (+ dW, dgamma,
    // dbeta), softmax VJP, dq = scale d(dots) k.  d(dots) leaves the chip once (fp32, into dA) for the product that needs every query tile of an image:
    // dK = scale d(dots)^T q; dV = attn'^T dO reads the normalised scores the forward kept.
    {
      const double pts = (double)b * h * a.nq * a.nk;
      Prof pr(e, "attn_deepvit_fused_bwd", 4.0 * pts * dh + 4.0 * pts * h, ((double)b * a.nq * 3 * h * dh + 2.0 * (double)b * a.nq * h * dh) * 2 + 8.0 * pts);
      launch_deepvit_attn_bwd((const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, a.ldq, a.ldk, a.ldv, a.qb, a.kb, a.vb, (const bf16_t*)gr.d_o, gr.ldo,
                              gr.ob, sc[0], e->params + bp.re_w, e->params + bp.re_g, dA, (bf16_t*)gr.dq, gr.lddq, gr.dqb, e->red_ws, e->grads + bp.re_w,
                              e->grads + bp.re_g, e->grads + bp.re_b, b, h, a.nq, a.nk, ld, scale, e->cfg.ln_eps, (const bf16_t*)e->zero_page, e->stream,
                              keep->sc_a2_bf16 ? 1 : 0);
    }
    // fp32 [nq][ld] score planes, or (kept as / written as bf16) [nq][ld2] planes with ld2 = nk rounded up to 8
    const int lp = keep->sc_a2_bf16 ? 1 : 0;
    const int64_t ldp = lp ? round_up(a.nk, 8) : ld, hsp = (int64_t)a.nq * ldp, bsp = (int64_t)h * hsp;
    bgemm_pair(e, BgemmCall{sc[pi], lp, 1, ldp, bsp, hsp, gr.d_o, T, gr.ldo, 1, gr.ob, dh, a.nk, dh, a.nq, b, h, EPI_STORE, T, gr.dv, gr.lddv, gr.dvb, dh, 1.0f},
               BgemmCall{dA, lp, 1, ldp, bsp, hsp, a.q, T, a.ldq, 1, a.qb, dh, a.nk, dh, a.nq, b, h, EPI_STORE, T, gr.dk, gr.lddk, gr.dkb, dh, scale});
    return;
  }
  // d(attn) = dO v^T ; dV = attn^T dO
  const int64_t ld2 = round_up(a.nk, 8), hs2 = (int64_t)a.nq * ld2, bs2 = (int64_t)h * hs2;   // bf16 score planes (a2_lp, ds_lp)
  bgemm_pair(e, BgemmCall{gr.d_o, T, gr.ldo, 1, gr.ob, dh, a.v, T, 1, a.ldv, a.vb, dh, a.nq, a.nk, dh, b, h, EPI_STORE_F32, 0, dA, ld, bs, hs, 1.0f},
             a2_lp ? BgemmCall{sc[pi], 1, 1, ld2, bs2, hs2, gr.d_o, T, gr.ldo, 1, gr.ob, dh, a.nk, dh, a.nq, b, h, EPI_STORE, T, gr.dv, gr.lddv, gr.dvb, dh, 1.0f}
                   : BgemmCall{sc[pi], 0, 1, ld, bs, hs, gr.d_o, T, gr.ldo, 1, gr.ob, dh, a.nk, dh, a.nq, b, h, EPI_STORE, T, gr.dv, gr.lddv, gr.dvb, dh, 1.0f});
  // (round 5, CaiT) d(dots) as bf16 into the buffer of the mixed softmax, which the dV product above was the last to read
  bf16_t* ds_lp = nullptr;
  // (the row-per-wave DeepViT chain reads the mixed scores; tensors kept by the one-kernel forward go to the point kernel, which recomputes them)
  const bool chain = !e->unfused_headops && headchain_supported(h, a.nk) && !(kept && keep->sc_no_mixed);
  if (chain && e->cfg.variant == VITX_VARIANT_CAIT) {
    Prof pr(e, "attn_headchain", 0, 0);
    if (T && a2_lp && e->score_bf16 && cait_chain_bwd_bf16_out_ok()) {
      ds_lp = (bf16_t*)sc[pi];
      //  d(dots) overwrites the forward's kept mixed softmax: a SECOND backward on the same forward must not read it as attention
      // weights -- the kept tensors are marked stale and that backward recomputes them (same bits as the forward's)
      if (kept) keep->sc_geom = -1;
    }
    launch_cait_chain_bwd(sc[0], sc[1], dA, e->params + bp.mix_pre, e->params + bp.mix_post, e->red_ws, e->grads + bp.mix_pre,
                          e->grads + bp.mix_post, b, h, a.nq, a.nk, ld, e->stream, ds_lp);
  } else if (chain && e->cfg.variant == VITX_VARIANT_DEEPVIT) {
    Prof pr(e, "attn_headchain", 0, 0);
    launch_deepvit_chain_bwd(sc[0], sc[1], dA, e->params + bp.re_w, e->params + bp.re_g, e->red_ws, e->grads + bp.re_w, e->grads + bp.re_g,
                             e->grads + bp.re_b, b, h, a.nq, a.nk, ld, e->cfg.ln_eps, e->stream);
  } else if (!e->unfused_headops && e->cfg.variant == VITX_VARIANT_DEEPVIT && deepvit_point_fwd_supported(h, a.nk)) {
    Prof pr(e, "attn_headchain", 0, 0);     // LayerNorm-over-heads VJP + mix VJP in one point kernel, then the softmax VJP
    launch_deepvit_point_bwd(sc[0], dA, e->params + bp.re_w, e->params + bp.re_g, e->red_ws, e->grads + bp.re_w,
                             e->grads + bp.re_g, e->grads + bp.re_b, b, h, a.nq, a.nk, ld, e->cfg.ln_eps, e->stream);
  } else {
    Prof pr(e, "attn_generic_headops", 0, 0);
    if (e->cfg.variant == VITX_VARIANT_CAIT) {
      launch_headmix_bwd(sc[1], dA, e->params + bp.mix_post, dA, e->red_ws, e->grads + bp.mix_post, b, h, a.nq, a.nk, ld, e->stream);
      launch_softmax_bwd_rows(sc[1], dA, rows, a.nk, ld, e->stream);
      launch_headmix_bwd(sc[0], dA, e->params + bp.mix_pre, dA, e->red_ws, e->grads + bp.mix_pre, b, h, a.nq, a.nk, ld, e->stream);
    } else if (e->cfg.variant == VITX_VARIANT_DEEPVIT) {
      launch_headnorm_bwd(sc[1], dA, e->params + bp.re_g, dA, e->red_ws, e->grads + bp.re_g, e->grads + bp.re_b, b, h, a.nq, a.nk, ld,
                          e->cfg.ln_eps, e->stream);
      launch_headmix_bwd(sc[0], dA, e->params + bp.re_w, dA, e->red_ws, e->grads + bp.re_w, b, h, a.nq, a.nk, ld, e->stream);
      launch_softmax_bwd_rows(sc[0], dA, rows, a.nk, ld, e->stream);
    } else {
      launch_softmax_bwd_rows(sc[0], dA, rows, a.nk, ld, e->stream);
    }
  }
  // dQ = scale dS k ; dK = scale dS^T q
  if (ds_lp) {
    bgemm_pair(e, BgemmCall{ds_lp, 1, ld2, 1, bs2, hs2, a.k, T, a.ldk, 1, a.kb, dh, a.nq, dh, a.nk, b, h, EPI_STORE, T, gr.dq, gr.lddq, gr.dqb, dh, scale},
               BgemmCall{ds_lp, 1, 1, ld2, bs2, hs2, a.q, T, a.ldq, 1, a.qb, dh, a.nk, dh, a.nq, b, h, EPI_STORE, T, gr.dk, gr.lddk, gr.dkb, dh, scale});
    return;
  }
  bgemm_pair(e, BgemmCall{dA, 0, ld, 1, bs, hs, a.k, T, a.ldk, 1, a.kb, dh, a.nq, dh, a.nk, b, h, EPI_STORE, T, gr.dq, gr.lddq, gr.dqb, dh, scale},
             BgemmCall{dA, 0, 1, ld, bs, hs, a.q, T, a.ldq, 1, a.qb, dh, a.nk, dh, a.nq, b, h, EPI_STORE, T, gr.dk, gr.lddk, gr.dkb, dh, scale});
}

static bool use_fused_attn(const vitx_engine* e, int n) {
  return e->bf16 && e->cfg.variant == VITX_VARIANT_VIT && !e->force_generic_attn && attn_bf16_supported(n, e->cfg.dim_head);
}
// BF16X3 mode: the fused split-operand attention (attn_x3.hip); VITX_X3_ATTN=2 keeps the materialised path with split-operand products, 0 exact ones
static bool use_fused_attn_x3(const vitx_engine* e, int n) {
  return e->x3 && e->x3_attn && e->x3_fused_attn && e->cfg.variant == VITX_VARIANT_VIT && !e->force_generic_attn && attn_x3_supported(n, e->cfg.dim_head);
}

// ------------------------------------------------------------------------------------------------
// one transformer block: x = attn(LN(x)) [*scale] + x ; x = mlp(LN(x)) [*scale] + x
//   (vit.py:99-104, deepvit.py:106-110, cait.py:146-153)
// ------------------------------------------------------------------------------------------------
static int block_forward(vitx_engine* e, Stage& st, int si, int l, int b, int nq, int nc, const float* context, float drop, uint64_t seed,
                         std::string& err) {
  const uint32_t site0 = (uint32_t)((si * 1000 + l) * 4);
  const BlockParams& bp = st.bp[l];
  BlockActs& ba = st.ba[l];
  const vitx_config& c = e->cfg;
  const int d = c.dim, inner = e->inner, m = c.mlp_dim, T = e->bf16, esz = e->esz;
  const int rows = b * nq;
  const int nk = nq + nc;
  const double lnb = (double)rows * d * (4 + esz);
  if (!ba.skip_attn) {
  {
    Prof pr(e, "layernorm_fwd", 0, lnb);
    launch_layernorm_fwd(ba.ln1_src ? ba.ln1_src : ba.x_in, d, e->params + bp.ln1_g, e->params + bp.ln1_b, ba.y1, T, d, ba.mean1, ba.rstd1, rows, d, c.ln_eps,
                         e->stream);
  }
  AttnView av;
  av.nq = nq; av.nk = nk; av.o = ba.o; av.ldo = inner; av.ob = (int64_t)nq * inner;
  if (c.variant == VITX_VARIANT_CAIT && bp.qkvcat.wt) {
    EpiParams ep; ep.out = ba.qkv; ep.ldo = 3 * inner;
    dense_fwd(e, ba.y1, d, rows, bp.qkvcat, EPI_STORE, ep);                  // cait.py:114-115 with context = x: q and kv of the same tokens, one launch
    av.q = ba.qkv; av.k = boff(ba.qkv, inner, esz); av.v = boff(ba.qkv, 2 * inner, esz);
    av.ldq = av.ldk = av.ldv = 3 * inner;
    av.qb = av.kb = av.vb = (int64_t)nq * 3 * inner;
  } else if (c.variant == VITX_VARIANT_CAIT) {
    EpiParams ep; ep.out = ba.q; ep.ldo = inner;
    dense_fwd(e, ba.y1, d, rows, bp.q, EPI_STORE, ep);                       // cait.py:114
    const void* ctx = ba.y1;
    if (nc > 0) {                                                            // cait.py:109-112
      Prof pr(e, "concat_ctx", 0, 0);
      launch_concat_ctx(ba.y1, T, context, ba.ctx, T, b, nq, nc, d, e->stream);
      ctx = ba.ctx;
    }
    EpiParams ep2; ep2.out = ba.kv; ep2.ldo = 2 * inner;
    dense_fwd(e, ctx, d, b * nk, bp.kv, EPI_STORE, ep2);                      // cait.py:115
    av.q = ba.q; av.ldq = inner; av.qb = (int64_t)nq * inner;
    av.k = ba.kv; av.ldk = 2 * inner; av.kb = (int64_t)nk * 2 * inner;
    av.v = boff(ba.kv, inner, esz); av.ldv = 2 * inner; av.vb = av.kb;       // cait.py:116
  } else {
    EpiParams ep; ep.out = ba.qkv; ep.ldo = 3 * inner;
    dense_fwd(e, ba.y1, d, rows, bp.qkv, EPI_STORE, ep);                     // vit.py:72
    av.q = ba.qkv; av.k = boff(ba.qkv, inner, esz); av.v = boff(ba.qkv, 2 * inner, esz);   // vit.py:73
    av.ldq = av.ldk = av.ldv = 3 * inner;
    av.qb = av.kb = av.vb = (int64_t)nq * 3 * inner;
  }
  if (use_fused_attn_x3(e, nq)) {
    Prof pr(e, "attn_x3_fwd", 4.0 * b * c.heads * (double)nq * nq * c.dim_head, (double)rows * inner * 4 * esz);
    launch_attn_x3_fwd((const float*)ba.qkv, (float*)ba.o, ba.lse, b, nq, c.heads, 1.0f / std::sqrt((float)c.dim_head), e->stream);
  } else if (use_fused_attn(e, nq)) {
    Prof pr(e, "attn_bf16_fwd", 4.0 * b * c.heads * (double)nq * nq * c.dim_head, (double)rows * inner * 4 * esz);
    launch_attn_bf16_fwd((const bf16_t*)ba.qkv, (bf16_t*)ba.o, ba.lse, b, nq, c.heads, 1.0f / std::sqrt((float)c.dim_head), (const bf16_t*)e->zero_page, (e->reverse_mask >> 2) & 1, e->stream);
  } else {
    const int need = c.variant == VITX_VARIANT_VIT ? 1 : 3;
    if (e->keep_scores && ba.sc_keep_elems == 0) {   // first use: this block's own score buffers, sized for the largest call
      const int64_t emax = (int64_t)c.max_batch * c.heads * st.nq_max * round_up(st.nq_max + st.nc_max, 4);
      if (e->sc_keep_bytes + need * emax * 4 <= e->sc_keep_budget) {
        for (int i = 0; i < need; ++i) DALLOC(ba.sc_keep[i], (size_t)emax * 4, false);
        ba.sc_keep_elems = emax;
        e->sc_keep_bytes += need * emax * 4;
      } else {
        ba.sc_keep_elems = -1;
      }
    }
    const int64_t elems = (int64_t)b * c.heads * nq * round_up(nk, 4);
    const bool keep = e->keep_scores && ba.sc_keep_elems >= elems;
    if (!keep) {
      ba.sc_geom = -1;
      int rc = ensure_scores(e, need, elems, err);
      if (rc != VITX_OK) return rc;
    }
    attn_generic_fwd(e, bp, av, b, keep ? &ba : nullptr);
  }
  if (bp.has_out && drop > 0.f) {
    // Dense -> Dropout -> (LayerScale) -> + residual, unfused (vit.py:61-69,83,101)
    EpiParams ep; ep.out = e->tmp_f32; ep.ldo = d;
    dense_fwd(e, ba.o, inner, rows, bp.out, EPI_STORE_F32, ep);
    Prof pr(e, "dropout", 0, 0);
    launch_dropout(e->tmp_f32, 0, (int64_t)rows * d, drop, seed, site0 + 1, e->stream);
    launch_axpy_resid(ba.x_in, e->tmp_f32, bp.a_scale >= 0 ? e->params + bp.a_scale : nullptr, ba.x_mid, bp.a_scale >= 0 ? ba.fa : nullptr, T,
                      rows, d, e->stream);
  } else if (bp.has_out) {
    EpiParams ep;
    ep.out = ba.x_mid; ep.ldo = d; ep.resid = ba.x_in; ep.ldr = d;
    if (bp.a_scale >= 0) { ep.scale = e->params + bp.a_scale; ep.out2 = ba.fa; ep.ldo2 = d; }
    dense_fwd(e, ba.o, inner, rows, bp.out, EPI_BIAS_RESID, ep);             // vit.py:83,101
  } else {
    Prof pr(e, "resid_add", 0, 0);
    launch_resid_add(ba.x_in, ba.o, T, ba.x_mid, (int64_t)rows * d, e->stream);   // vit.py:53 (to_out is identity)
  }
  }   // !skip_attn
  if (ba.skip_mlp) return VITX_OK;
  {
    Prof pr(e, "layernorm_fwd", 0, lnb);
    launch_layernorm_fwd(ba.ln2_src ? ba.ln2_src : ba.x_mid, d, e->params + bp.ln2_g, e->params + bp.ln2_b, ba.y2, T, d, ba.mean2, ba.rstd2, rows, d, c.ln_eps,
                         e->stream);
  }
  {
    EpiParams ep; ep.out = ba.hpre; ep.ldo = m; ep.out2 = ba.act; ep.ldo2 = m;
    ep.nt_out = e->nt_mask & 1;
    dense_fwd(e, ba.y2, d, rows, bp.fc1, EPI_BIAS_GELU, ep);                  // vit.py:39,34
  }
  if (drop > 0.f) {
    {
      Prof pr(e, "dropout", 0, 0);
      launch_dropout(ba.act, T, (int64_t)rows * m, drop, seed, site0 + 2, e->stream);     // vit.py:41
    }
    EpiParams ep; ep.out = e->tmp_f32; ep.ldo = d;
    dense_fwd(e, ba.act, m, rows, bp.fc2, EPI_STORE_F32, ep);
    Prof pr(e, "dropout", 0, 0);
    launch_dropout(e->tmp_f32, 0, (int64_t)rows * d, drop, seed, site0 + 3, e->stream);   // vit.py:43
    launch_axpy_resid(ba.x_mid, e->tmp_f32, bp.m_scale >= 0 ? e->params + bp.m_scale : nullptr, ba.x_out, bp.m_scale >= 0 ? ba.fm : nullptr, T,
                      rows, d, e->stream);
  } else {
    EpiParams ep;
    ep.out = ba.x_out; ep.ldo = d; ep.resid = ba.x_mid; ep.ldr = d;
    if (bp.m_scale >= 0) { ep.scale = e->params + bp.m_scale; ep.out2 = ba.fm; ep.ldo2 = d; }
    dense_fwd(e, ba.act, m, rows, bp.fc2, EPI_BIAS_RESID, ep);                // vit.py:42,102
  }
  return VITX_OK;
}

// g (fp32 [rows,d]) holds dL/dx_out on entry and dL/dx_in on exit; g_lp is its T copy (bf16 mode).
static inline int64_t branch_key(int si, int l, int mlp) { return (((int64_t)si + 1) << 32) | ((int64_t)l << 1) | mlp; }
// next_l: the layer of the same stage (same token rows) whose backward runs right after this one, or -1 -- lets this block's last LayerNorm VJP carry
// the LayerScale VJP of that layer's MLP branch (block_layernorm_bwd, NextBranch)
static int block_backward(vitx_engine* e, Stage& st, int si, int l, int b, int nq, int nc, float drop, uint64_t seed, std::string& err, int next_l = -1) {
  const uint32_t site0 = (uint32_t)((si * 1000 + l) * 4);
  const BlockParams& bp = st.bp[l];
  BlockActs& ba = st.ba[l];
  const vitx_config& c = e->cfg;
  const int d = c.dim, inner = e->inner, m = c.mlp_dim, T = e->bf16, esz = e->esz;
  const int rows = b * nq, nk = nq + nc;
  const void* gT = T ? e->g_lp : (const void*)e->g;   // T view of the residual gradient (re-read after each LayerNorm VJP: side_rotate)
  const double lnb = (double)rows * d * (4 + 4 + 4 + esz + esz);
  // Parallel half-blocks (parallel_vit.py:36-42): every branch of a group sees the SAME output gradient g, and the gradient of the
  // group's common LayerNorm input is g + sum_i LN_i^T(...): it is accumulated in g2 while g / g_lp keep feeding the remaining
  // branches, and becomes the new g (and g_lp) with the last branch processed.
  const bool grouped = ba.par_first || ba.par_last || ba.ln1_src || ba.ln2_src;
  const float* ln_gin = (!grouped || ba.par_first) ? e->g : e->g2;
  float* ln_gout = (!grouped || ba.par_last) ? e->g : e->g2;
  // (round 5) every branch with a LayerScale takes its input gradient from the scale VJP pass (d_br), never from g_lp: when all blocks have one
  // (CaiT) the LayerNorm VJPs do not write the bf16 copy at all -- 2 bytes per element of the residual stream per VJP, 52 per cfg5 step
  bool glp_dead = e->glp_skip;
  for (const Stage& sx : e->stages)
    for (const BlockParams& bx : sx.bp) glp_dead = glp_dead && bx.a_scale >= 0 && bx.m_scale >= 0;
  const bool ln_writes_glp = T && (!grouped || ba.par_last) && !glp_dead;
  // the bf16 residual gradient a LayerNorm VJP writes: the next ring slot while weight gradients on the side stream may still read the current one
  auto next_glp = [&]() -> void* {
    if (!ln_writes_glp) return nullptr;
    side_rotate(e, e->rg_glp, e->g_lp);
    return e->g_lp;
  };

  const void* dbranch = gT;
  bool fc2_bias_done = false, out_bias_done = false;   // bias gradient already produced by the LayerScale VJP pass
  const int64_t key_mlp = branch_key(si, l, 1), key_attn = branch_key(si, l, 0);
  if (!ba.skip_mlp) {
  // ---- MLP branch: x_out = x_mid + scale * fc2(gelu(fc1(LN(x_mid))))
  if (e->dbr_ready == key_mlp) {           // the LayerNorm VJP of the layer above has already run this branch's LayerScale VJP (block_layernorm_bwd, nb)
    e->dbr_ready = 0;
    dbranch = e->d_br;
    fc2_bias_done = dense_gb(e, bp.fc2) != nullptr;
  } else if (bp.m_scale >= 0 || drop > 0.f) {   // LayerScale VJP (cait.py:47-48): dscale = sum g*f(x), d f = g*scale; Dropout VJP: same mask
    Prof pr(e, "branch_grad", 0, 0);
    // LayerScale without dropout: one pass over g gives dscale AND the branch gradient g * scale
    const bool one_pass = bp.m_scale >= 0 && drop == 0.f && d % 4 == 0;
    fc2_bias_done = one_pass && dense_gb(e, bp.fc2) != nullptr;   // ... and the fc2 bias gradient = column sums of that branch gradient
    side_rotate(e, e->rg_dbr, e->d_br);
    if (bp.m_scale >= 0)
      launch_scale_grad(ba.fm, T, d, e->g, d, rows, d, e->red_ws, e->grads + bp.m_scale, e->stream, one_pass ? e->params + bp.m_scale : nullptr,
                        one_pass ? e->d_br : nullptr, d, fc2_bias_done ? dense_gb(e, bp.fc2) : nullptr);
    if (!one_pass) launch_branch_grad(e->g, bp.m_scale >= 0 ? e->params + bp.m_scale : nullptr, e->d_br, T, rows, d, drop, seed, site0 + 3, e->stream);
    dbranch = e->d_br;
  }
  // fc1 bias gradient = column sums of d hpre: produced per M-tile by the fc2-dgrad epilogue itself (no second pass over the
  // 310 MB output) whenever that GEMM runs an LDS-staged bf16 kernel and no dropout mask is applied to d hpre afterwards
  const bool fc1_bias_fused = e->bf16 && !e->force_generic_gemm && !(e->gemm_kernel & 256) && drop == 0.f && bp.fc1.b >= 0;
  const int cs_rows = (int)ceil_div(rows, 96);    // >= the number of wave rows of every variant (96 / 128 / 160 rows per wave row, 128 per tile of the lockstep kernels)
  const hipEvent_t fork_fc2 = side_prefork(e);    // act and the branch gradient are complete here: the fc2 weight gradient may start
  hipEvent_t fork_fc1 = nullptr;
  {
    side_rotate(e, e->rg_dh, e->d_h);
    EpiParams ep; ep.out = e->d_h; ep.ldo = m; ep.aux = ba.hpre; ep.ldaux = m;
    // with the side stream live the per-tile sums go to a ring slot of their own and their reduction to the small-reduction stream (as the
    // LayerNorm VJP's: block_layernorm_bwd)
    const bool cs_side = fc1_bias_fused && e->side_live && e->side2 && e->rg_cs.n && rows >= e->side2_min_rows;
    if (cs_side) side_rotate(e, e->rg_cs, e->cs_part);
    float* cs = cs_side ? e->cs_part : e->red_ws;
    if (fc1_bias_fused) {
      // rows the chosen tile shape does not reach stay 0.  (round 6: our own fill kernel, not hipMemsetAsync -- the runtime's blit path cost the chain a
      //  22-us gap in front of every one of these 6-us fills, 12 per step: profiles/r6/trace_streams_main_queue_gaps_r6i.log)
      launch_fill_zero(cs, (int64_t)cs_rows * m * 4, e->stream);
      ep.colsum = cs; ep.ldcs = m;
    }
    dense_dgrad(e, dbranch, d, rows, bp.fc2, EPI_GELU_BWD, ep);             // d hpre = (d act) * gelu'(hpre)
    if (fc1_bias_fused) {
      Prof pr(e, "reduce_partials", 0, (double)(cs_rows + 1) * m * 4);
      if (cs_side) side2_begin(e);
      launch_reduce_partials3(cs, cs_rows, m, m, 1, e->grads + bp.fc1.b, nullptr, nullptr, cs + (int64_t)cs_rows * m, 1.0f, cs_side ? e->side2 : e->stream);
      if (cs_side) side2_end(e, e->rg_cs);   // (zeroing the slot here instead of on the chain: measured neutral, r4z)
    }
    if (drop > 0.f) {
      Prof pr(e, "dropout", 0, 0);
      launch_dropout(e->d_h, T, (int64_t)rows * m, drop, seed, site0 + 2, e->stream);   // mask of the post-GELU dropout (commutes)
    }
    fork_fc1 = side_prefork(e);                   // d hpre is complete: the fc1 weight gradient need not wait for the fc1 input gradient
  }
  // Order: the two consumers of d hpre (310 MB at ViT-B/16, just written) run right behind its producer; the fc2 weight gradient,
  // which reads other tensors (act, the branch gradient), follows them instead of sitting in between and pushing d hpre out of the
  // memory-side cache.  VITX_MLP_BWD_ORDER=0: fc2 weight gradient first (the order of round 1; same results either way).
  const bool fc2_bias_in_ln = dbranch != e->d_br && !grouped;   // db_fc2 = column sums of g: fused into the LayerNorm backward pass below
  auto fc2_param_grads = [&]() {
    dense_wgrad(e, ba.act, m, dbranch, d, rows, bp.fc2, fork_fc2);
    side_note_read(e, dbranch == e->d_br ? e->rg_dbr : e->rg_glp);
    if (!fc2_bias_in_ln && !fc2_bias_done) bias_grad(e, dbranch == e->d_br ? (const void*)e->d_br : (const void*)e->g, dbranch == e->d_br ? T : 0, d, rows, bp.fc2);
  };
  if (!e->mlp_bwd_consumers_first) fc2_param_grads();
  {
    EpiParams ep; ep.out = e->d_y; ep.ldo = d;
    ep.nt_out = (e->nt_mask >> 4) & 1;   // d(y2) is read by the LayerNorm backward only after both weight gradients: keep d(hpre) cached instead
    dense_dgrad(e, e->d_h, m, rows, bp.fc1, EPI_STORE, ep);
  }
  dense_wgrad(e, ba.y2, d, e->d_h, m, rows, bp.fc1, fork_fc1);
  side_note_read(e, e->rg_dh);
  if (!fc1_bias_fused) bias_grad(e, e->d_h, T, m, rows, bp.fc1);
  if (e->mlp_bwd_consumers_first) fc2_param_grads();
  {
    Prof pr(e, "layernorm_bwd", 0, lnb);
    // the attention branch of this block is next: its LayerScale VJP on this pass where it has one (and no dropout mask to apply)
    NextBranch nb{ba.fa, bp.a_scale >= 0 ? e->params + bp.a_scale : nullptr, bp.a_scale >= 0 ? e->grads + bp.a_scale : nullptr,
                  (bp.has_out && dense_gb(e, bp.out)) ? dense_gb(e, bp.out) : nullptr};
    const bool fuse = e->ln_scale_fused && T && !grouped && !ba.skip_attn && bp.a_scale >= 0 && (bp.has_out ? drop : 0.f) == 0.f && !fc2_bias_in_ln &&
                      layernorm_bwd_scale_ok(d);
    void* ln_glp = fuse ? nullptr : next_glp();
    if (fuse) { side_rotate(e, e->rg_dbr, e->d_br); e->dbr_ready = key_attn; }
    block_layernorm_bwd(e, e->d_y, T, d, ba.ln2_src ? ba.ln2_src : ba.x_mid, ba.mean2, ba.rstd2, e->params + bp.ln2_g, ln_gin, ln_gout, ln_glp,
                        e->grads + bp.ln2_g, e->grads + bp.ln2_b, fc2_bias_in_ln ? e->grads + bp.fc2.b : nullptr, rows, fuse ? &nb : nullptr);
  }
  }   // !skip_mlp
  if (ba.skip_attn) {
    report_ready(e, bp.p_begin, bp.p_end - bp.p_begin);
    return VITX_OK;
  }

  // ---- attention branch: x_mid = x_in + scale * to_out(attn(LN(x_in)))
  gT = T ? e->g_lp : (const void*)e->g;
  dbranch = gT;
  if (e->dbr_ready == key_attn) {
    e->dbr_ready = 0;
    dbranch = e->d_br;
    out_bias_done = bp.has_out && dense_gb(e, bp.out) != nullptr;
  } else if (bp.a_scale >= 0 || (drop > 0.f && bp.has_out)) {
    Prof pr(e, "branch_grad", 0, 0);
    const float adrop = bp.has_out ? drop : 0.f;
    const bool one_pass = bp.a_scale >= 0 && adrop == 0.f && d % 4 == 0;
    out_bias_done = one_pass && bp.has_out && dense_gb(e, bp.out) != nullptr;
    side_rotate(e, e->rg_dbr, e->d_br);
    if (bp.a_scale >= 0)
      launch_scale_grad(ba.fa, T, d, e->g, d, rows, d, e->red_ws, e->grads + bp.a_scale, e->stream, one_pass ? e->params + bp.a_scale : nullptr,
                        one_pass ? e->d_br : nullptr, d, out_bias_done ? dense_gb(e, bp.out) : nullptr);
    if (!one_pass) launch_branch_grad(e->g, bp.a_scale >= 0 ? e->params + bp.a_scale : nullptr, e->d_br, T, rows, d, adrop, seed, site0 + 1, e->stream);
    dbranch = e->d_br;
  }
  bool out_bias_in_ln = false;
  const void* d_o = dbranch;   // when to_out is the identity (vit.py:53) the branch gradient IS d(attn_out)
  if (bp.has_out) {
    EpiParams ep; ep.out = e->d_o; ep.ldo = inner;
    const hipEvent_t fork_out = side_prefork(e);   // o and the branch gradient are complete
    dense_dgrad(e, dbranch, d, rows, bp.out, EPI_STORE, ep);
    dense_wgrad(e, ba.o, inner, dbranch, d, rows, bp.out, fork_out);
    side_note_read(e, dbranch == e->d_br ? e->rg_dbr : e->rg_glp);
    out_bias_in_ln = dbranch != e->d_br && !grouped;
    if (!out_bias_in_ln && !out_bias_done) bias_grad(e, dbranch == e->d_br ? (const void*)e->d_br : (const void*)e->g, dbranch == e->d_br ? T : 0, d, rows, bp.out);
    d_o = e->d_o;
  }
  side_rotate(e, e->rg_dqkv, e->d_qkv);   // d(q, k, v) of this block: the previous block's may still feed its weight gradient
  AttnView av;
  av.nq = nq; av.nk = nk; av.o = ba.o; av.ldo = inner; av.ob = (int64_t)nq * inner;
  AttnGrad ag;
  ag.d_o = d_o; ag.ldo = inner; ag.ob = (int64_t)nq * inner;
  if (c.variant == VITX_VARIANT_CAIT && bp.qkvcat.wt) {
    // patch stage, concatenated [to_q | to_kv] operands: d(q | k | v) rows as one matrix -> ONE input-gradient GEMM straight into d(y1) (no second
    // GEMM + add pass), two weight gradients reading their column blocks of it
    av.q = ba.qkv; av.k = boff(ba.qkv, inner, esz); av.v = boff(ba.qkv, 2 * inner, esz);
    av.ldq = av.ldk = av.ldv = 3 * inner;
    av.qb = av.kb = av.vb = (int64_t)nq * 3 * inner;
    ag.dq = e->d_qkv; ag.dk = boff(e->d_qkv, inner, esz); ag.dv = boff(e->d_qkv, 2 * inner, esz);
    ag.lddq = ag.lddk = ag.lddv = 3 * inner;
    ag.dqb = ag.dkb = ag.dvb = (int64_t)nq * 3 * inner;
    int rc = ensure_scores(e, 4, (int64_t)b * c.heads * nq * round_up(nk, 4), err);
    if (rc != VITX_OK) return rc;
    attn_generic_bwd(e, bp, av, ag, b, &ba);
    EpiParams ep; ep.out = e->d_y; ep.ldo = d;
    const hipEvent_t fork_qkv = side_prefork(e);                                // d(q | k | v) is complete
    dense_dgrad(e, e->d_qkv, 3 * inner, rows, bp.qkvcat, EPI_STORE, ep);
    dense_wgrad(e, ba.y1, d, ag.dq, 3 * inner, rows, bp.q, fork_qkv);
    dense_wgrad(e, ba.y1, d, ag.dk, 3 * inner, rows, bp.kv, fork_qkv);
    side_note_read(e, e->rg_dqkv);
  } else if (c.variant == VITX_VARIANT_CAIT) {
    av.q = ba.q; av.ldq = inner; av.qb = (int64_t)nq * inner;
    av.k = ba.kv; av.ldk = 2 * inner; av.kb = (int64_t)nk * 2 * inner;
    av.v = boff(ba.kv, inner, esz); av.ldv = 2 * inner; av.vb = av.kb;
    // d_qkv buffer reused as [dq | dkv]: dq [rows, inner], then dkv [b*nk, 2*inner]
    void* dq = e->d_qkv;
    void* dkv = boff(e->d_qkv, (round_up((int64_t)rows, 256) + 320) * inner, esz);
    ag.dq = dq; ag.lddq = inner; ag.dqb = (int64_t)nq * inner;
    ag.dk = dkv; ag.lddk = 2 * inner; ag.dkb = (int64_t)nk * 2 * inner;
    ag.dv = boff(dkv, inner, esz); ag.lddv = 2 * inner; ag.dvb = ag.dkb;
    int rc = ensure_scores(e, 4, (int64_t)b * c.heads * nq * round_up(nk, 4), err);
    if (rc != VITX_OK) return rc;
    attn_generic_bwd(e, bp, av, ag, b, &ba);
    // to_q / to_kv VJPs
    const void* ctx = nc > 0 ? ba.ctx : ba.y1;
    EpiParams ep; ep.out = e->d_y; ep.ldo = d;
    const hipEvent_t fork_qkv = side_prefork(e);                                // dq and dkv are complete
    dense_dgrad(e, dq, inner, rows, bp.q, EPI_STORE, ep);                       // d y1 (via q)
    dense_wgrad(e, ba.y1, d, dq, inner, rows, bp.q, fork_qkv);
    EpiParams ep2; ep2.out = e->d_ctx; ep2.ldo = d;
    dense_dgrad(e, dkv, 2 * inner, b * nk, bp.kv, EPI_STORE, ep2);              // d ctx (via k, v)
    dense_wgrad(e, ctx, d, dkv, 2 * inner, b * nk, bp.kv, fork_qkv);
    side_note_read(e, e->rg_dqkv);
    Prof pr(e, "ctx_bwd", 0, 0);
    if (nc > 0) {
      side_rotate(e, e->rg_dbr, e->d_br);
      // d ctx = [d y1 part | d context part]  (cait.py:109-112 VJP); context is the un-normalised patch output
      launch_split_ctx_bwd(e->d_ctx, T, e->d_br, e->g_ctx, b, nq, nc, d, e->stream);
      launch_add_T(e->d_y, e->d_br, T, (int64_t)rows * d, e->stream);
    } else {
      launch_add_T(e->d_y, e->d_ctx, T, (int64_t)rows * d, e->stream);
    }
  } else {
    av.q = ba.qkv; av.k = boff(ba.qkv, inner, esz); av.v = boff(ba.qkv, 2 * inner, esz);
    av.ldq = av.ldk = av.ldv = 3 * inner;
    av.qb = av.kb = av.vb = (int64_t)nq * 3 * inner;
    ag.dq = e->d_qkv; ag.dk = boff(e->d_qkv, inner, esz); ag.dv = boff(e->d_qkv, 2 * inner, esz);
    ag.lddq = ag.lddk = ag.lddv = 3 * inner;
    ag.dqb = ag.dkb = ag.dvb = (int64_t)nq * 3 * inner;
    if (use_fused_attn_x3(e, nq)) {
      Prof pr(e, "attn_x3_bwd", 14.0 * b * c.heads * (double)nq * nq * c.dim_head, (double)rows * inner * 8 * esz);
      launch_attn_x3_bwd((const float*)ba.qkv, (const float*)ba.o, (const float*)d_o, ba.lse, (float*)e->d_qkv, b, nq, c.heads,
                         1.0f / std::sqrt((float)c.dim_head), e->stream);
    } else if (use_fused_attn(e, nq)) {
      Prof pr(e, "attn_bf16_bwd", 14.0 * b * c.heads * (double)nq * nq * c.dim_head, (double)rows * inner * 8 * esz);
      launch_attn_bf16_bwd((const bf16_t*)ba.qkv, (const bf16_t*)ba.o, (const bf16_t*)d_o, ba.lse, e->dsum, (bf16_t*)e->d_qkv, b, nq,
                           c.heads, 1.0f / std::sqrt((float)c.dim_head), (const bf16_t*)e->zero_page, e->stream);
    } else {
      int rc = ensure_scores(e, 4, (int64_t)b * c.heads * nq * round_up(nk, 4), err);
      if (rc != VITX_OK) return rc;
      attn_generic_bwd(e, bp, av, ag, b, &ba);
    }
    EpiParams ep; ep.out = e->d_y; ep.ldo = d;
    ep.nt_out = (e->nt_mask >> 4) & 1;
    const hipEvent_t fork_qkv = side_prefork(e);   // d(q, k, v) is complete
    dense_dgrad(e, e->d_qkv, 3 * inner, rows, bp.qkv, EPI_STORE, ep);
    dense_wgrad(e, ba.y1, d, e->d_qkv, 3 * inner, rows, bp.qkv, fork_qkv);
    side_note_read(e, e->rg_dqkv);
  }
  {
    Prof pr(e, "layernorm_bwd", 0, lnb);
    // the MLP branch of the layer below is next
    NextBranch nb{nullptr, nullptr, nullptr, nullptr};
    bool fuse = false;
    if (next_l >= 0 && e->ln_scale_fused && T && !grouped && drop == 0.f && !out_bias_in_ln && layernorm_bwd_scale_ok(d)) {
      const BlockParams& np_ = st.bp[(size_t)next_l];
      const BlockActs& na = st.ba[(size_t)next_l];
      if (np_.m_scale >= 0 && !na.skip_mlp && !(na.par_first || na.par_last || na.ln1_src || na.ln2_src)) {
        nb = NextBranch{na.fm, e->params + np_.m_scale, e->grads + np_.m_scale, dense_gb(e, np_.fc2)};
        fuse = true;
      }
    }
    void* ln_glp = fuse ? nullptr : next_glp();
    if (fuse) { side_rotate(e, e->rg_dbr, e->d_br); e->dbr_ready = branch_key(si, next_l, 1); }
    block_layernorm_bwd(e, e->d_y, T, d, ba.ln1_src ? ba.ln1_src : ba.x_in, ba.mean1, ba.rstd1, e->params + bp.ln1_g, ln_gin, ln_gout, ln_glp,
                        e->grads + bp.ln1_g, e->grads + bp.ln1_b, out_bias_in_ln ? e->grads + bp.out.b : nullptr, rows, fuse ? &nb : nullptr);
  }
  report_ready(e, bp.p_begin, bp.p_end - bp.p_begin);
  return VITX_OK;
}

// ------------------------------------------------------------------------------------------------
// creation
// ------------------------------------------------------------------------------------------------
static bool env_flag(const char* n) { return vitx_env_flag(n); }

// body of engine_create: on any failure the caller (engine_create) releases everything `e` owns so far
void engine_destroy(vitx_engine* e);
static int engine_create_body(vitx_engine* e, const vitx_config& cfg, std::string& err) {
  e->cfg = cfg;
  if (e->cfg.ln_eps <= 0.f) e->cfg.ln_eps = 1e-3f;
  if (e->cfg.channels <= 0) e->cfg.channels = 3;
  if (e->cfg.max_batch <= 0) e->cfg.max_batch = 1;
  const vitx_config& c = e->cfg;
  std::string perr = build_param_table(c, e->table);
  if (!perr.empty()) { err = perr; return VITX_ERR_INVALID; }
  e->n_params = e->table.back().offset + e->table.back().count;
  e->n_arena = e->table.back().aoff + round_up(e->table.back().count, 4);
  if (c.compute != VITX_COMPUTE_FP32_PARITY && c.compute != VITX_COMPUTE_BF16 && c.compute != VITX_COMPUTE_BF16X3) { err = "unknown compute mode"; return VITX_ERR_INVALID; }
  e->bf16 = c.compute == VITX_COMPUTE_BF16;
  e->x3 = c.compute == VITX_COMPUTE_BF16X3;
  e->x3_attn = true;   // the materialised attention products too (8.3 vs 9.2 ms per ViT-B/16 step at batch 64); VITX_X3_ATTN=0 keeps them exact
  if (const char* k = vitx_env("VITX_X3_ATTN")) { e->x3_attn = atoi(k) != 0; e->x3_fused_attn = atoi(k) == 1; }
  e->esz = e->bf16 ? 2 : 4;
  e->inner = c.heads * c.dim_head;
  e->np_max = (c.image_h / c.patch_h) * (c.image_w / c.patch_w);
  const bool cait = c.variant == VITX_VARIANT_CAIT;
  e->ntok_max = cait ? e->np_max : e->np_max + 1;        // rows of pos_embedding (the merger ViT keeps np + 1 rows and uses np)
  e->ntok_cap = cait ? e->ntok_max : e->ntok_max + 1;
  e->pd = c.patch_h * c.patch_w * c.channels;
  e->pd_k = (int)round_up(e->pd, 64);
  e->nc_k = (int)round_up(c.num_classes, 64);
  if (c.dim > 4096) { err = "dim must be <= 4096"; return VITX_ERR_UNSUPPORTED; }   // (any width: rows that are not multiples of 4 floats take the scalar-tail kernels)
  if (c.heads > 32 && c.variant != VITX_VARIANT_VIT) { err = "heads > 32 unsupported for DeepViT/CaiT"; return VITX_ERR_UNSUPPORTED; }
  if (e->bf16 && (c.dim % 64 || e->inner % 64 || c.mlp_dim % 64)) {
    err = "BF16 compute needs dim, heads*dim_head and mlp_dim to be multiples of 64 (use FP32_PARITY otherwise)";
    return VITX_ERR_UNSUPPORTED;
  }
  if (cait && c.cls_depth < 0) { err = "cls_depth must be >= 0"; return VITX_ERR_INVALID; }
  e->force_generic_gemm = env_flag("VITX_GENERIC_GEMM");
  e->force_generic_attn = env_flag("VITX_GENERIC_ATTN");
  if (const char* k = vitx_env("VITX_DEEPVIT_FUSED")) e->deepvit_fused = atoi(k) != 0;
  if (const char* k = vitx_env("VITX_DEEPVIT_FUSED_BWD")) e->deepvit_fused_bwd = atoi(k) != 0;
  if (const char* k = vitx_env("VITX_CAIT_FUSED")) e->cait_fused = atoi(k) != 0;
  if (const char* k = vitx_env("VITX_CAIT_QKV_CAT")) e->cait_qkv_cat = atoi(k) != 0;
  if (const char* k = vitx_env("VITX_GLP_SKIP")) e->glp_skip = atoi(k) != 0;
  if (const char* k = vitx_env("VITX_SCORE_BF16")) e->score_bf16 = atoi(k) != 0;
  if (const char* k = vitx_env("VITX_LN_SCALE_FUSED")) e->ln_scale_fused = atoi(k) != 0;
  if (const char* k = vitx_env("VITX_MLP_BWD_ORDER")) e->mlp_bwd_consumers_first = atoi(k) != 0;
  if (const char* k = vitx_env("VITX_NT")) e->nt_mask = atoi(k);
  if (const char* k = vitx_env("VITX_BGEMM_PAIRS")) e->bgemm_pairs = atoi(k) != 0;
  if (const char* k = vitx_env("VITX_REVERSE")) e->reverse_mask = atoi(k);
  gemm_f32_mfma_read_env();
  if (const char* k = vitx_env("VITX_REVERSE_MIN_MB")) e->reverse_min_bytes = (int64_t)atoi(k) << 20;
  if (const char* k = vitx_env("VITX_GEMM_KERNEL")) e->gemm_kernel = atoi(k);
  if (const char* k = vitx_env("VITX_GEMM_TAIL_KERNEL")) e->gemm_tail = atoi(k);   // with VITX_GEMM_KERNEL=13 / 11: tile variant of the tail launch (1, 3, 10; gemm_bf16.hip, tail balancing)
  if (const char* k = vitx_env("VITX_UNFUSED_HEADOPS")) e->unfused_headops = atoi(k) != 0;
  e->wgrad_via_transpose = env_flag("VITX_WGRAD_TRANSPOSE");
  e->keep_scores = !env_flag("VITX_RECOMPUTE_SCORES");
  if (const char* k = vitx_env("VITX_SC_KEEP_MB")) e->sc_keep_budget = (int64_t)atoll(k) << 20;   // budget of the kept score tensors (blocks beyond it recompute)
  if (const char* k = vitx_env("VITX_GEMM_STAGGER")) {
    e->gemm_stagger = atoi(k);
    if (e->gemm_stagger & 3) fprintf(stderr, "[vitx] VITX_GEMM_STAGGER=%d: timing experiment bits set -- GEMM results are WRONG in this process\n", e->gemm_stagger);
  }

  HIPCHK(hipSetDevice(c.device_id));
  HIPCHK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
  e->stream = e->own_stream;
  if (const char* k = vitx_env("VITX_SIDE_STREAM")) e->side_mode = atoi(k);
  if (const char* k = vitx_env("VITX_LN_REDUCE_SIDE_ROWS")) e->side2_min_rows = atoi(k);
  if (e->bf16 && e->side_mode) {
    // lowest priority: the input-gradient chain is the critical path, the weight gradients fill what it leaves free (VITX_SIDE_STREAM=2: same priority)
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    HIPCHK(hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, e->side_mode == 2 ? greatest : least));
    HIPCHK(hipStreamCreateWithPriority(&e->side2, hipStreamNonBlocking, greatest));
  }
  // (round 6) The handle's FOUR streams -- compute, weight gradients, small reductions, communication -- are created HERE, back to back, and each is
  // made to submit something at once.  The driver spreads a process's hardware queues over the GPU's four command-processor pipes in creation order, and
  // two queues on one pipe are served by one micro-engine: a communication queue whose head is a barrier packet (waiting, for a millisecond, for the
  // weight gradients of the block it will send) stalls the dispatches of a compute queue behind it on the same pipe.  Created late (at
  // vitx_comm_overlap, after whatever streams torch / RCCL had made in between) the communication stream landed on the compute stream's pipe or not
  // by luck: 43.5 ms per ViT-B/16 step instead of 35.0 with a collective that does NOTHING, 48 ms with real RCCL and VITX_SIDE_STREAM=0
  // (profiles/r6/stream_to_pipe_mapping_r6.md).  Four consecutive queues are four different pipes, whatever was created before them.
  {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    static const int prio_env = [] { const char* v = vitx_env("VITX_COMM_PRIORITY"); return v ? atoi(v) : 1; }();   // 1 highest (default), 0 default priority, -1 lowest
    static const int early_env = [] { const char* v = vitx_env("VITX_COMM_STREAM_EARLY"); return v ? atoi(v) : 1; }();   // 0: created at vitx_comm_overlap (A/B)
    if (early_env) {
      HIPCHK(hipStreamCreateWithPriority(&e->cm.stream, hipStreamNonBlocking, prio_env > 0 ? greatest : (prio_env < 0 ? least : 0)));
      void* probe = nullptr;
      HIPCHK(hipMalloc(&probe, 256));
      e->allocs.push_back(probe);
      for (hipStream_t st : {e->own_stream, e->side, e->side2, e->cm.stream})   // first submission = the hardware queue exists, in this order
        if (st) launch_fill_zero(probe, 64, st);
      for (hipStream_t st : {e->own_stream, e->side, e->side2, e->cm.stream})
        if (st) HIPCHK(hipStreamSynchronize(st));
    }
  }

  const int d = c.dim, inner = e->inner, m = c.mlp_dim, esz = e->esz;
  const int64_t B = c.max_batch;
  // row padding: a multiple of 256 plus 320 spare rows (the 320-row GEMM tile may cover up to 319 rows past M)
  e->mp = round_up(B * e->ntok_cap, 256) + 320;
  e->mpp = round_up(B * e->np_max, 256) + 320;
  e->bp = round_up(B, 256) + 320;

  DALLOC(e->params, (size_t)e->n_arena * 4, false);
  DALLOC(e->grads, (size_t)e->n_arena * 4, false);

  e->pos = find_param(e, "pos_embedding");
  e->cls = find_param(e, "cls_token");
  e->head_g = find_param(e, "mlp_head.norm.gamma");
  e->head_b = find_param(e, "mlp_head.norm.beta");
  int rc;
  if ((rc = init_dense(e, e->patch, "patch_embedding.kernel", "patch_embedding.bias", e->pd, d, err)) != VITX_OK) return rc;
  if ((rc = init_dense(e, e->head, "mlp_head.kernel", "mlp_head.bias", d, c.num_classes, err)) != VITX_OK) return rc;

  // stages
  auto make_stage = [&](const std::string& prefix, int depth, int nq, int nc) -> int {
    Stage st;
    st.prefix = prefix; st.depth = depth; st.nq_max = nq; st.nc_max = nc;
    st.bp.resize(depth);
    st.ba.resize(depth);
    const int64_t rows = round_up(B * nq, 256) + 320, crow = round_up(B * (nq + nc), 256) + 320;
    float* x_prev = nullptr;
    for (int l = 0; l < depth; ++l) {
      BlockParams& bp = st.bp[l];
      BlockActs& ba = st.ba[l];
      const std::string pre = prefix + "." + std::to_string(l);
      bp.a_scale = find_param(e, pre + ".attn.scale");
      bp.ln1_g = find_param(e, pre + ".attn.norm.gamma");
      bp.ln1_b = find_param(e, pre + ".attn.norm.beta");
      if (cait) {
        // patch stage (no context tokens) in bf16 mode: one Dense on the concatenated operand copies of to_q and to_kv (build_convert_table)
        const bool cat = e->bf16 && nc == 0 && e->cait_qkv_cat && inner % 64 == 0 && !e->force_generic_gemm;   // (the fp32-FMA debug path reads the parameter tensors themselves)
        if ((rc = init_dense(e, bp.q, pre + ".attn.to_q.kernel", "", d, inner, err, !cat)) != VITX_OK) return rc;
        if ((rc = init_dense(e, bp.kv, pre + ".attn.to_kv.kernel", "", d, 2 * inner, err, !cat)) != VITX_OK) return rc;
        if (cat) {
          Dense& w = bp.qkvcat;
          w.in = d; w.out = 3 * inner; w.in_k = (int)round_up(d, 64); w.out_k = (int)round_up(3 * inner, 64);
          // NOT t-buffers (round 6; found by tools/fuzz_configs.py "sequences"): ensure_geometry re-zeroes every registered T buffer when the batch
          // or image size of a handle changes, and these are WEIGHTS -- registered (as they were since round 5) they were wiped by the first call
          // with another geometry and stayed zero until the next parameter update refreshed them: a bf16 CaiT handle evaluated at a second batch
          // size computed its patch-stage attention from q = k = v = 0 (logits off by ~1e-2, every gradient through to_q / to_kv zero).
          DALLOC(w.wt, (size_t)round_up(w.out, 256) * w.in_k * 2, false);
          DALLOC(w.wn, (size_t)round_up(w.in, 256) * w.out_k * 2, false);
        }
        bp.mix_pre = find_param(e, pre + ".attn.mix_heads_pre_attn");
        bp.mix_post = find_param(e, pre + ".attn.mix_heads_post_attn");
      } else {
        if ((rc = init_dense(e, bp.qkv, pre + ".attn.to_qkv.kernel", "", d, 3 * inner, err)) != VITX_OK) return rc;
      }
      bp.re_w = find_param(e, pre + ".attn.reattn_weights");
      bp.re_g = find_param(e, pre + ".attn.reattn_norm.gamma");
      bp.re_b = find_param(e, pre + ".attn.reattn_norm.beta");
      bp.has_out = find_param(e, pre + ".attn.to_out.kernel") >= 0;
      if (bp.has_out && (rc = init_dense(e, bp.out, pre + ".attn.to_out.kernel", pre + ".attn.to_out.bias", inner, d, err)) != VITX_OK) return rc;
      if (!bp.has_out && inner != d) { err = "to_out missing but inner_dim != dim"; return VITX_ERR_INVALID; }
      bp.m_scale = find_param(e, pre + ".mlp.scale");
      bp.ln2_g = find_param(e, pre + ".mlp.norm.gamma");
      bp.ln2_b = find_param(e, pre + ".mlp.norm.beta");
      if ((rc = init_dense(e, bp.fc1, pre + ".mlp.fc1.kernel", pre + ".mlp.fc1.bias", d, m, err)) != VITX_OK) return rc;
      if ((rc = init_dense(e, bp.fc2, pre + ".mlp.fc2.kernel", pre + ".mlp.fc2.bias", m, d, err)) != VITX_OK) return rc;
      bp.p_begin = cait ? bp.a_scale : bp.ln1_g;
      bp.p_end = bp.fc2.b + d;
      // activations
      if (x_prev) ba.x_in = x_prev; else DALLOC(ba.x_in, (size_t)rows * d * 4, false);
      DALLOC(ba.x_mid, (size_t)rows * d * 4, false);
      DALLOC(ba.x_out, (size_t)rows * d * 4, false);
      x_prev = ba.x_out;
      DALLOC(ba.y1, (size_t)rows * d * esz, true);
      if (cait && bp.qkvcat.wt) {
        DALLOC(ba.qkv, (size_t)rows * 3 * inner * esz, true);   // [q | k | v] per token row, as vit.py's to_qkv would lay them out
        DALLOC(ba.fa, (size_t)rows * d * esz, true);
        DALLOC(ba.fm, (size_t)rows * d * esz, true);
      } else if (cait) {
        DALLOC(ba.q, (size_t)rows * inner * esz, true);
        DALLOC(ba.kv, (size_t)crow * 2 * inner * esz, true);
        if (nc > 0) DALLOC(ba.ctx, (size_t)crow * d * esz, true);
        DALLOC(ba.fa, (size_t)rows * d * esz, true);
        DALLOC(ba.fm, (size_t)rows * d * esz, true);
      } else {
        DALLOC(ba.qkv, (size_t)rows * 3 * inner * esz, true);
      }
      DALLOC(ba.o, (size_t)rows * inner * esz, true);
      DALLOC(ba.y2, (size_t)rows * d * esz, true);
      DALLOC(ba.hpre, (size_t)rows * m * esz, true);
      DALLOC(ba.act, (size_t)rows * m * esz, true);
      DALLOC(ba.mean1, (size_t)rows * 4, false);
      DALLOC(ba.rstd1, (size_t)rows * 4, false);
      DALLOC(ba.mean2, (size_t)rows * 4, false);
      DALLOC(ba.rstd2, (size_t)rows * 4, false);
      DALLOC(ba.lse, (size_t)B * c.heads * nq * 4 + 16, false);
    }
    e->stages.push_back(std::move(st));
    return VITX_OK;
  };
  // parallel_vit.py:104-117: layer l = P attention half-blocks then P feed-forward half-blocks, all normalising the layer's input
  // and adding into one running residual.  Each half-block is a Stage entry with only its own half's parameters and activations.
  auto make_parallel_stage = [&](int depth, int P, int nq) -> int {
    Stage st;
    st.prefix = "transformer"; st.depth = depth * 2 * P; st.nq_max = nq; st.nc_max = 0;
    st.bp.resize(st.depth);
    st.ba.resize(st.depth);
    const int64_t rows = round_up(B * nq, 256) + 320;
    float* x_prev = nullptr;
    if (st.depth > 0) DALLOC(x_prev, (size_t)rows * d * 4, false);
    for (int l = 0; l < depth; ++l) {
      float* x_layer = x_prev;                     // what the attention branches normalise
      for (int i = 0; i < P; ++i) {
        BlockParams& bp = st.bp[(size_t)(l * 2 * P + i)];
        BlockActs& ba = st.ba[(size_t)(l * 2 * P + i)];
        const std::string pre = "transformer." + std::to_string(l) + ".attn." + std::to_string(i);
        bp.ln1_g = find_param(e, pre + ".norm.gamma");
        bp.ln1_b = find_param(e, pre + ".norm.beta");
        if ((rc = init_dense(e, bp.qkv, pre + ".to_qkv.kernel", "", d, 3 * inner, err)) != VITX_OK) return rc;
        bp.has_out = find_param(e, pre + ".to_out.kernel") >= 0;
        if (bp.has_out && (rc = init_dense(e, bp.out, pre + ".to_out.kernel", pre + ".to_out.bias", inner, d, err)) != VITX_OK) return rc;
        if (!bp.has_out && inner != d) { err = "to_out missing but inner_dim != dim"; return VITX_ERR_INVALID; }
        bp.p_begin = bp.ln1_g;
        bp.p_end = bp.has_out ? bp.out.b + d : bp.qkv.w + (int64_t)d * 3 * inner;
        ba.skip_mlp = true;
        ba.x_in = x_prev;
        ba.ln1_src = i == 0 ? nullptr : x_layer;
        DALLOC(ba.x_mid, (size_t)rows * d * 4, false);
        ba.x_out = ba.x_mid;
        x_prev = ba.x_mid;
        ba.par_last = i == 0; ba.par_first = i == P - 1;   // the backward walks the half-blocks in reverse
        DALLOC(ba.y1, (size_t)rows * d * esz, true);
        DALLOC(ba.qkv, (size_t)rows * 3 * inner * esz, true);
        DALLOC(ba.o, (size_t)rows * inner * esz, true);
        DALLOC(ba.mean1, (size_t)rows * 4, false);
        DALLOC(ba.rstd1, (size_t)rows * 4, false);
        DALLOC(ba.lse, (size_t)B * c.heads * nq * 4 + 16, false);
      }
      x_layer = x_prev;                            // what the feed-forward branches normalise
      for (int i = 0; i < P; ++i) {
        BlockParams& bp = st.bp[(size_t)(l * 2 * P + P + i)];
        BlockActs& ba = st.ba[(size_t)(l * 2 * P + P + i)];
        const std::string pre = "transformer." + std::to_string(l) + ".mlp." + std::to_string(i);
        bp.ln2_g = find_param(e, pre + ".norm.gamma");
        bp.ln2_b = find_param(e, pre + ".norm.beta");
        if ((rc = init_dense(e, bp.fc1, pre + ".fc1.kernel", pre + ".fc1.bias", d, m, err)) != VITX_OK) return rc;
        if ((rc = init_dense(e, bp.fc2, pre + ".fc2.kernel", pre + ".fc2.bias", m, d, err)) != VITX_OK) return rc;
        bp.p_begin = bp.ln2_g;
        bp.p_end = bp.fc2.b + d;
        ba.skip_attn = true;
        ba.x_in = x_prev; ba.x_mid = x_prev;
        ba.ln2_src = i == 0 ? nullptr : x_layer;
        DALLOC(ba.x_out, (size_t)rows * d * 4, false);
        x_prev = ba.x_out;
        ba.par_last = i == 0; ba.par_first = i == P - 1;
        DALLOC(ba.y2, (size_t)rows * d * esz, true);
        DALLOC(ba.hpre, (size_t)rows * m * esz, true);
        DALLOC(ba.act, (size_t)rows * m * esz, true);
        DALLOC(ba.mean2, (size_t)rows * 4, false);
        DALLOC(ba.rstd2, (size_t)rows * 4, false);
      }
    }
    e->stages.push_back(std::move(st));
    return VITX_OK;
  };
  const bool merger = c.variant == VITX_VARIANT_PATCH_MERGER;
  const int nbranch = c.num_parallel_branches > 1 ? c.num_parallel_branches : 1;
  if (cait) {
    if ((rc = make_stage("patch_transformer", c.depth, e->np_max, 0)) != VITX_OK) return rc;
    if ((rc = make_stage("cls_transformer", c.cls_depth, 1, e->np_max)) != VITX_OK) return rc;
  } else if (nbranch > 1) {
    if ((rc = make_parallel_stage(c.depth, nbranch, e->ntok_cap)) != VITX_OK) return rc;
  } else {
    if ((rc = make_stage("transformer", c.depth, e->ntok_cap, 0)) != VITX_OK) return rc;
  }
  if (merger) {
    e->cfg.pool = VITX_POOL_MEAN;                                          // Reduce('b n d -> b d', 'mean')  vit_with_patch_merger.py:169
    e->merge_t = c.patch_merge_num_tokens;
    e->merge_after = (c.patch_merge_layer > 0 ? c.patch_merge_layer : c.depth / 2) - 1;   // default(patch_merge_layer, depth // 2) - 1  :117
    if (e->merge_after >= c.depth) e->merge_after = -1;                   // `index == patch_merge_layer_index` never holds (:131)
    e->pm_g = find_param(e, "transformer.patch_merger.norm.gamma");
    e->pm_b = find_param(e, "transformer.patch_merger.norm.beta");
    e->pm_q = find_param(e, "transformer.patch_merger.queries");
    if (e->merge_after >= 0) {
      const int64_t rn = round_up(B * e->np_max, 256) + 320, rt = round_up(B * e->merge_t, 256) + 320;
      DALLOC(e->pm_xn, (size_t)rn * d * 4, false);
      DALLOC(e->pm_mean, (size_t)rn * 4, false);
      DALLOC(e->pm_rstd, (size_t)rn * 4, false);
      DALLOC(e->pm_attn, (size_t)B * e->merge_t * round_up(e->np_max, 4) * 4, false);
      DALLOC(e->pm_dattn, (size_t)B * e->merge_t * round_up(e->np_max, 4) * 4, false);
      DALLOC(e->pm_out, (size_t)rt * d * 4, false);
      DALLOC(e->pm_dxn, (size_t)rn * d * 4, false);
      DALLOC(e->pm_dxn2, (size_t)rn * d * 4, false);
      DALLOC(e->pm_dq, (size_t)rt * d * 4, false);
      Stage& st = e->stages[0];
      if (e->merge_after + 1 < st.depth) st.ba[(size_t)e->merge_after + 1].x_in = e->pm_out;   // the next layer reads the merged tokens
    }
  }

  // shared buffers
  const int64_t crow_max = cait ? round_up(B * (1 + e->np_max), 256) + 320 : e->mp;
  const int64_t rmax = std::max(e->mp, crow_max);
  DALLOC(e->img_dev, (size_t)B * c.image_h * c.image_w * c.channels * 4, false);
  // (NOT a t-buffer: its row padding is kept by prepare_patch_rows, which every writer calls.  Registered -- as it was until round 6 -- it was wiped by
  //  the ensure_geometry of a LATER call of the same logical step: MAE / SimMIM run patch_tokens_forward [writes e->patches] and then
  //  transformer_forward on another geometry (b, visible tokens), so from the second batch size on the patch-embedding weight gradient of a bf16
  //  wrapper was computed from zeroed patches = exactly zero.  tests/test_gpu_wrappers.py::test_wrappers_follow_a_changing_batch_on_one_object)
  DALLOC(e->patches, (size_t)e->mpp * e->pd_k * esz, false);
  DALLOC(e->pooled, (size_t)e->bp * d * 4, false);
  DALLOC(e->yh, (size_t)e->bp * d * esz, true);
  DALLOC(e->mean_h, (size_t)e->bp * 4, false);
  DALLOC(e->rstd_h, (size_t)e->bp * 4, false);
  DALLOC(e->logits, (size_t)e->bp * e->nc_k * 4, false);
  DALLOC(e->dlogits, (size_t)e->bp * e->nc_k * 4, false);
  DALLOC(e->dl_lp, (size_t)e->bp * e->nc_k * esz, true);
  DALLOC(e->dyh, (size_t)e->bp * d * esz, true);
  DALLOC(e->dpooled, (size_t)e->bp * d * 4, false);
  DALLOC(e->loss_rows, (size_t)e->bp * 4, false);
  DALLOC(e->g, (size_t)rmax * d * 4, false);
  if (c.num_parallel_branches > 1) DALLOC(e->g2, (size_t)rmax * d * 4, false);
  if (e->bf16) DALLOC(e->g_lp, (size_t)rmax * d * esz, true);
  if (cait) DALLOC(e->g_ctx, (size_t)e->mpp * d * 4, false);
  DALLOC(e->d_h, (size_t)rmax * m * esz, true);
  DALLOC(e->d_y, (size_t)rmax * d * esz, true);
  DALLOC(e->d_o, (size_t)rmax * inner * esz, true);
  DALLOC(e->d_qkv, (size_t)(rmax + 256) * 3 * inner * esz + (size_t)crow_max * 2 * inner * esz, true);
  DALLOC(e->d_ctx, (size_t)crow_max * d * esz, true);
  DALLOC(e->d_br, (size_t)rmax * d * esz, true);
  if (e->side) {
    // ring slots of the buffers the input-gradient chain rewrites while side-stream weight gradients read them (slot 0 = the buffer above)
    auto ring = [&](SideRing& r, void* first, size_t bytes, int n) -> int {
      r.n = n; r.cur = 0; r.slot[0] = first;
      for (int i = 1; i < n; ++i) DALLOC(r.slot[i], bytes, true);
      for (int i = 0; i < n; ++i) HIPCHK(hipEventCreateWithFlags(&r.rd[i], hipEventDisableTiming));
      return VITX_OK;
    };
    if ((rc = ring(e->rg_dh, e->d_h, (size_t)rmax * m * esz, 2)) != VITX_OK) return rc;
    if ((rc = ring(e->rg_glp, e->g_lp, (size_t)rmax * d * esz, 3)) != VITX_OK) return rc;
    if ((rc = ring(e->rg_dqkv, e->d_qkv, (size_t)(rmax + 256) * 3 * inner * esz + (size_t)crow_max * 2 * inner * esz, 2)) != VITX_OK) return rc;
    if ((rc = ring(e->rg_dbr, e->d_br, (size_t)rmax * d * esz, 3)) != VITX_OK) return rc;
    DALLOC(e->ln_part, (size_t)layernorm_bwd_ws_elems(d) * 4, false);
    // (one slot per use of a backward pass -- no wait for these on the chain at all -- measured the same as four: r4pg)
    if ((rc = ring(e->rg_lnp, e->ln_part, (size_t)layernorm_bwd_ws_elems(d) * 4, 4)) != VITX_OK) return rc;
    const size_t cs_bytes = (size_t)(ceil_div(rmax, 96) + 64) * m * 4;   // per-tile column sums + the second reduction level behind them
    DALLOC(e->cs_part, cs_bytes, false);
    if ((rc = ring(e->rg_cs, e->cs_part, cs_bytes, 4)) != VITX_OK) return rc;
  }
  DALLOC(e->dsum, (size_t)B * c.heads * e->ntok_cap * 4 + 16, false);
  DALLOC(e->zero_page, 256, false);
  DALLOC(e->tmp_f32, (size_t)rmax * std::max<int64_t>(d, e->pd) * 4, false);
  const int64_t maxfeat = std::max<int64_t>({(int64_t)d, 3LL * inner, (int64_t)m, (int64_t)e->pd_k, (int64_t)e->nc_k});
  if (e->x3) {   // split-K partials of the weight gradients (dense_wgrad): at most 32 slices of the largest kernel, never more than 256 MB
    const int64_t max_w = std::max<int64_t>({(int64_t)d * 3 * inner, (int64_t)d * m, (int64_t)e->pd * d, (int64_t)d * c.num_classes, (int64_t)inner * d});
    e->partial_elems = std::min<int64_t>(64LL * 1024 * 1024, 32 * max_w);
    DALLOC(e->partial_ws, (size_t)e->partial_elems * 4, false);
  }
  if (e->bf16) {
    e->t_rows = round_up(maxfeat, 256);
    if (e->wgrad_via_transpose) {
      DALLOC(e->xt, (size_t)e->t_rows * rmax * 2, false);
      DALLOC(e->dyt, (size_t)e->t_rows * rmax * 2, false);
    }
    const int64_t max_w = std::max<int64_t>({(int64_t)d * 3 * inner, (int64_t)d * m, (int64_t)e->pd * d, (int64_t)d * c.num_classes, (int64_t)inner * d});
    e->partial_elems = 512LL * 256 * 256 + 2 * max_w;
    DALLOC(e->partial_ws, (size_t)e->partial_elems * 4, false);
  }
  // (+ per-M-tile column sums of the fc2-dgrad epilogue: one row per 256 token rows, 32 second-level rows)
  e->red_elems = std::max<int64_t>({layernorm_bwd_ws_elems(d), colsum_ws_elems((int)maxfeat), headmix_ws_elems((int)B, c.heads, 1, 1),
                                    (int64_t)256 * 2 * 32, (int64_t)(1024 + 64) * d, (ceil_div(std::max<int64_t>(e->mp, e->mpp), 96) + 40) * (int64_t)m,
                                    headchain_ws_elems(c.heads), deepvit_point_ws_elems((int)B, c.heads, e->ntok_cap),
                                    deepvit_point_bwd_ws_elems(c.heads), deepvit_attn_bwd_ws_elems((int)B, c.heads, e->ntok_cap)});
  DALLOC(e->red_ws, (size_t)e->red_elems * 4, false);
  HIPCHK(hipStreamSynchronize(e->stream));
  
  return VITX_OK;
}

void engine_destroy(vitx_engine* e) {
  if (!e) return;
  (void)hipStreamSynchronize(e->stream);
  if (e->side) (void)hipStreamSynchronize(e->side);
  if (e->side2) (void)hipStreamSynchronize(e->side2);
  comm_destroy(e);
  for (void* p : e->allocs) (void)hipFree(p);
  for (SideRing* r : {&e->rg_dh, &e->rg_glp, &e->rg_dqkv, &e->rg_dbr, &e->rg_lnp, &e->rg_cs})
    for (int i = 0; i < SIDE_RING_MAX; ++i) if (r->rd[i]) (void)hipEventDestroy(r->rd[i]);
  for (auto ev : e->side_events) (void)hipEventDestroy(ev);
  if (e->side) (void)hipStreamDestroy(e->side);
  if (e->side2) (void)hipStreamDestroy(e->side2);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
}

int engine_create(const vitx_config& cfg, vitx_engine** out, std::string& err) {
  vitx_engine* e = new vitx_engine();
  const int rc = engine_create_body(e, cfg, err);
  if (rc != VITX_OK) {          // a failed hipMalloc half-way (e.g. out of memory at a larger max_batch) must not strand the buffers,
    engine_destroy(e);          // the stream and the object allocated before it
    return rc;
  }
  build_convert_table(e);
  *out = e;
  return VITX_OK;
}

// rows >= M of every T buffer must be zero (they are K-padding of the wgrad GEMMs): re-zero when the geometry changes
static void ensure_geometry(vitx_engine* e, int b, int ntok) {
  const int64_t geom = ((int64_t)b << 32) | (uint32_t)ntok;
  if (geom == e->zero_geom) return;
  if (e->zero_geom >= 0 && e->bf16)
    for (auto& tb : e->t_buffers) (void)hipMemsetAsync(tb.first, 0, tb.second, e->stream);
  e->zero_geom = geom;
}

// e->patches is written by the full forward (geometry (b, ntok)) and by patch_tokens_forward (its own geometry): keep "rows beyond
// the last unfold are zero" (K padding of the patch-embedding weight gradient) whichever of the two ran before
static void prepare_patch_rows(vitx_engine* e, int64_t rows) {
  if (e->patch_rows >= 0 && e->patch_rows != rows && e->bf16)
    (void)hipMemsetAsync(e->patches, 0, (size_t)e->mpp * e->pd_k * e->esz, e->stream);
  e->patch_rows = rows;
  e->have_pt = false;
  e->have_embed = false;       // whoever overwrites e->patches re-establishes its own state afterwards
}

// ------------------------------------------------------------------------------------------------
// PatchMerger (vit_with_patch_merger.py:42-55): x [b, n, d] -> LayerNorm -> softmax(queries @ xn^T * d^-0.5) @ xn -> [b, t, d].
// t is a handful of tokens: exact-fp32 batched GEMMs of the generic kernel in both compute modes (launch-bound, not FLOP-bound).
// ------------------------------------------------------------------------------------------------
static void merger_gemm(vitx_engine* e, const float* A, int64_t sam, int64_t sak, int64_t sAb, const float* B, int64_t sbk, int64_t sbn, int64_t sBb,
                        int M, int N, int K, int nb, float alpha, float* out, int64_t ldo, int64_t out_bstride, const float* resid) {
  GenericGemmArgs g;
  g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.nb = nb; g.sAb = sAb; g.sBb = sBb;
  EpiParams ep;
  ep.out = out; ep.ldo = ldo; ep.out_batch_stride = out_bstride; ep.M = M; ep.N = N; ep.alpha = alpha;
  (void)resid;
  finalize_epi(ep);
  Prof pr(e, "patch_merger", 2.0 * nb * M * (double)N * K, 0);
  launch_gemm_generic(g, ep, EPI_STORE_F32, 0, 0, 0, e->stream);
}

static void merger_forward(vitx_engine* e, const float* x, int b, int n) {
  const vitx_config& c = e->cfg;
  const int d = c.dim, t = e->merge_t;
  const int64_t ldn = round_up(n, 4);
  const float scale = 1.0f / std::sqrt((float)d);                                        // self.scale = dim ** -0.5   :45
  {
    Prof pr(e, "layernorm_fwd", 0, (double)b * n * d * 8);
    launch_layernorm_fwd(x, d, e->params + e->pm_g, e->params + e->pm_b, e->pm_xn, 0, d, e->pm_mean, e->pm_rstd, b * n, d, c.ln_eps, e->stream);   // :50
  }
  // sim = queries @ (xn^T * scale)   :51
  merger_gemm(e, e->params + e->pm_q, d, 1, 0, e->pm_xn, 1, d, (int64_t)n * d, t, n, d, b, scale, e->pm_attn, ldn, (int64_t)t * ldn, nullptr);
  {
    Prof pr(e, "softmax", 0, 0);
    launch_softmax_rows(e->pm_attn, (int64_t)b * t, n, ldn, e->stream);                  // :52
  }
  merger_gemm(e, e->pm_attn, ldn, 1, (int64_t)t * ldn, e->pm_xn, d, 1, (int64_t)n * d, t, d, n, b, 1.0f, e->pm_out, d, (int64_t)t * d, nullptr);   // :53
}

// g holds d(out) [b, t, d] on entry and d(x) [b, n, d] on exit (plus its T copy); queries / norm gradients go to the arena
static int merger_backward(vitx_engine* e, const float* x, int b, int n, std::string& err) {
  const vitx_config& c = e->cfg;
  const int d = c.dim, t = e->merge_t, T = e->bf16;
  const int64_t ldn = round_up(n, 4);
  const float scale = 1.0f / std::sqrt((float)d);
  float* dout = e->tmp_f32;                                                              // [b, t, d]: g is overwritten below
  HIPCHK(hipMemcpyAsync(dout, e->g, (size_t)b * t * d * 4, hipMemcpyDeviceToDevice, e->stream));
  // d attn = d out @ xn^T;  d xn = attn^T @ d out
  merger_gemm(e, dout, d, 1, (int64_t)t * d, e->pm_xn, 1, d, (int64_t)n * d, t, n, d, b, 1.0f, e->pm_dattn, ldn, (int64_t)t * ldn, nullptr);
  merger_gemm(e, e->pm_attn, 1, ldn, (int64_t)t * ldn, dout, d, 1, (int64_t)t * d, n, d, t, b, 1.0f, e->pm_dxn, d, (int64_t)n * d, nullptr);
  {
    Prof pr(e, "softmax_bwd", 0, 0);
    launch_softmax_bwd_rows(e->pm_attn, e->pm_dattn, (int64_t)b * t, n, ldn, e->stream);  // d sim (in place)
  }
  // d queries = scale * sum_b d sim[b] @ xn[b]: per image, then a fixed-order sum over the batch
  merger_gemm(e, e->pm_dattn, ldn, 1, (int64_t)t * ldn, e->pm_xn, d, 1, (int64_t)n * d, t, d, n, b, scale, e->pm_dq, d, (int64_t)t * d, nullptr);
  launch_batch_reduce(e->pm_dq, b, t, d, 0, t, e->grads + e->pm_q, e->stream);
  // d xn += scale * d sim^T @ queries
  merger_gemm(e, e->pm_dattn, 1, ldn, (int64_t)t * ldn, e->params + e->pm_q, d, 1, 0, n, d, t, b, scale, e->pm_dxn2, d, (int64_t)n * d, nullptr);
  launch_resid_add(e->pm_dxn, e->pm_dxn2, 0, e->pm_dxn2, (int64_t)b * n * d, e->stream);
  {
    Prof pr(e, "layernorm_bwd", 0, 0);
    launch_layernorm_bwd(e->pm_dxn2, 0, d, x, d, e->pm_mean, e->pm_rstd, e->params + e->pm_g, nullptr, 0, e->g, d, nullptr, 0, e->red_ws,
                         e->grads + e->pm_g, e->grads + e->pm_b, nullptr, b * n, d, e->stream);
    if (T) launch_convert(e->g, d, e->g_lp, 1, d, b * n, d, d, e->stream);   // the fp32-dy form of the kernel has no bf16 side output
  }
  return VITX_OK;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// cait.py:17-31: python-level random skipping of whole blocks (numpy RNG in the reference; NOT gated by `training`,
// cait.py:147).  Host-side, seeded by the call's seed; at least one layer survives per stage.
static void draw_layer_dropout(vitx_engine* e, uint64_t seed) {
  e->layer_kept.assign(e->stages.size(), {});
  uint64_t st = seed * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL;
  auto next = [&]() {
    st += 0x9E3779B97F4A7C15ULL;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  };
  for (size_t si = 0; si < e->stages.size(); ++si) {
    const int n = e->stages[si].depth;
    auto& kept = e->layer_kept[si];
    kept.assign((size_t)n, true);
    const float p = e->cfg.variant == VITX_VARIANT_CAIT ? e->cfg.layer_dropout : 0.f;
    if (p <= 0.f || n == 0) continue;
    int alive = 0;
    for (int i = 0; i < n; ++i) {
      kept[(size_t)i] = !((double)(next() >> 11) * (1.0 / 9007199254740992.0) < (double)p);
      alive += kept[(size_t)i];
    }
    if (alive == 0) kept[(size_t)(next() % (uint64_t)n)] = true;   // "make sure at least one layer makes it"
  }
}

// ntok rows per image in memory, of which the first ntok - extra are pooled (a distillation token is split off first, distill.py:32-33)
static int head_forward(vitx_engine* e, const float* x_last, int b, int ntok, int extra, float* logits_dev, std::string& err) {
  const vitx_config& c = e->cfg;
  e->have_head = false;        // the head's saved statistics are about to be overwritten (engine_head_forward sets it again)
  const int d = c.dim, T = e->bf16;
  const float* src = x_last;
  int64_t ldsrc = (int64_t)ntok * d;               // cls pooling: row 0 of every image (vit.py:173, cait.py:192)
  if (c.variant != VITX_VARIANT_CAIT && c.pool == VITX_POOL_MEAN) {   // vit.py:170-171
    Prof pr(e, "pool", 0, 0);
    launch_mean_pool(x_last, b, ntok - extra, d, e->pooled, e->stream, ntok);
    src = e->pooled;
    ldsrc = d;
  }
  {
    Prof pr(e, "layernorm_fwd", 0, 0);
    launch_layernorm_fwd(src, ldsrc, e->params + e->head_g, e->params + e->head_b, e->yh, T, d, e->mean_h, e->rstd_h, b, d, c.ln_eps, e->stream);
  }
  EpiParams ep; ep.out = e->logits; ep.ldo = e->nc_k;
  dense_fwd(e, e->yh, d, b, e->head, EPI_STORE_F32, ep);                       // vit.py:156
  if (logits_dev)
    HIPCHK(hipMemcpy2DAsync(logits_dev, (size_t)c.num_classes * 4, e->logits, (size_t)e->nc_k * 4, (size_t)c.num_classes * 4, b,
                            hipMemcpyDeviceToDevice, e->stream));
  return VITX_OK;
}

int engine_forward(vitx_engine* e, const float* img_dev, int b, int H, int W, int training, uint64_t seed, float* logits_dev,
                   std::string& err, const float* distill_token_dev, float* distill_out_dev) {
  const vitx_config& c = e->cfg;
  const int extra = distill_token_dev ? 1 : 0;
  if (extra && c.variant == VITX_VARIANT_CAIT) { err = "distillation token: ViT / DeepViT only"; return VITX_ERR_UNSUPPORTED; }
  if (b <= 0 || b > c.max_batch) { err = "batch must be in [1, max_batch]"; return VITX_ERR_INVALID; }
  // caller-supplied patch rows [b, np, pd] instead of an image (engine_forward_patches: T2T-ViT's tokenizer output feeding the
  // patch Dense, t2t.py:74-75,106): the unfold is skipped, everything else is the ordinary forward
  const float* patches_in = e->fwd_patches;
  e->fwd_patches = nullptr;
  if (patches_in) {
    if (e->fwd_np <= 0 || e->fwd_np > e->np_max) { err = "forward_patches: np must be in [1, num_patches]"; return VITX_ERR_INVALID; }
    H = e->fwd_np * c.patch_h; W = c.patch_w;      // np x 1 patches: keeps b * H * W * C == b * np * pd for the staging buffers
  } else if (H <= 0 || W <= 0 || H > c.image_h || W > c.image_w || H % c.patch_h || W % c.patch_w) {
    err = "Image dimensions must be divisible by the patch size.";            // and fit the configured pos_embedding (vit.py:165)
    return VITX_ERR_INVALID;
  }
  const bool cait = c.variant == VITX_VARIANT_CAIT, merger = c.variant == VITX_VARIANT_PATCH_MERGER;
  if (extra && merger) { err = "distillation token: ViT / DeepViT only"; return VITX_ERR_UNSUPPORTED; }
  const bool no_cls = cait || merger;                                   // cait.py:181-184; vit_with_patch_merger.py:173-177
  const int np = (H / c.patch_h) * (W / c.patch_w);
  const int ntok = no_cls ? np : np + 1 + extra;
  const int d = c.dim, T = e->bf16;
  ensure_geometry(e, b, ntok);
  if (e->params_dirty) engine_refresh_weights(e);
  Stage& s0 = e->stages[0];
  float* x0 = s0.depth > 0 ? s0.ba[0].x_in : e->tmp_f32;
  prepare_patch_rows(e, (int64_t)b * np);
  if (patches_in) {
    launch_convert(patches_in, e->pd, e->patches, T, e->pd_k, b * np, e->pd, e->pd_k, e->stream);
  } else {
    Prof pr(e, "patch_unfold", 0, (double)b * H * W * c.channels * 4 + (double)b * np * e->pd_k * e->esz);
    launch_unfold(img_dev, e->patches, T, b, H, W, c.channels, c.patch_h, c.patch_w, e->pd_k, e->stream);   // vit.py:142
  }
  e->last_from_patches = patches_in != nullptr;
  if (!no_cls) {
    Prof pr(e, "cls_pos_row", 0, 0);
    launch_cls_pos_row(x0, e->params + e->cls, e->params + e->pos, b, ntok, d, d, e->stream);              // vit.py:163-165
  }
  {
    EpiParams ep;
    ep.out = x0; ep.ldo = d; ep.pos = e->params + e->pos; ep.ldr = d;
    ep.np = np; ep.ntok = ntok; ep.tok_off = no_cls ? 0 : 1;
    dense_fwd(e, e->patches, e->pd_k, b * np, e->patch, EPI_PATCH, ep);                                     // vit.py:143 (+164-165)
  }
  if (extra)   // x = concat([x, distill_tokens], axis=1), after the position embedding (distill.py:24-28)
    launch_set_token_row(x0, distill_token_dev, b, ntok, ntok - 1, d, e->stream);
  const float drop = training ? c.dropout : 0.f, emb_drop = training ? c.emb_dropout : 0.f;
  if (emb_drop > 0.f) {
    Prof pr(e, "dropout", 0, 0);
    launch_dropout(x0, 0, (int64_t)b * ntok * d, emb_drop, seed, 0u, e->stream);                           // vit.py:166
  }
  draw_layer_dropout(e, seed);
  int rc;
  for (int l = 0; l < s0.depth; ++l) {
    if (!e->layer_kept[0][(size_t)l]) {   // cait.py:17-31,147: the whole block is skipped
      HIPCHK(hipMemcpyAsync(s0.ba[l].x_out, s0.ba[l].x_in, (size_t)b * ntok * d * 4, hipMemcpyDeviceToDevice, e->stream));
      continue;
    }
    const bool merged = merger && e->merge_after >= 0 && l > e->merge_after;
    if ((rc = block_forward(e, s0, 0, l, b, merged ? e->merge_t : ntok, 0, nullptr, drop, seed, err)) != VITX_OK) return rc;
    if (merger && l == e->merge_after) merger_forward(e, s0.ba[l].x_out, b, ntok);      // vit_with_patch_merger.py:131-132
  }
  const float* x_last = s0.depth > 0 ? s0.ba[s0.depth - 1].x_out : x0;
  int head_tok = ntok;
  if (merger && e->merge_after >= 0) {
    head_tok = e->merge_t;
    if (e->merge_after == s0.depth - 1) x_last = e->pm_out;
  }
  if (cait) {
    Stage& s1 = e->stages[1];
    float* xc = s1.depth > 0 ? s1.ba[0].x_in : e->pooled;
    {
      Prof pr(e, "cls_broadcast", 0, 0);
      launch_broadcast_rows(e->params + e->cls, d, xc, b, e->stream);                                        // cait.py:189
    }
    for (int l = 0; l < s1.depth; ++l) {
      if (!e->layer_kept[1][(size_t)l]) {
        HIPCHK(hipMemcpyAsync(s1.ba[l].x_out, s1.ba[l].x_in, (size_t)b * d * 4, hipMemcpyDeviceToDevice, e->stream));
        continue;
      }
      if ((rc = block_forward(e, s1, 1, l, b, 1, np, x_last, drop, seed, err)) != VITX_OK) return rc;       // cait.py:190
    }
    x_last = s1.depth > 0 ? s1.ba[s1.depth - 1].x_out : xc;
    head_tok = 1;
  }
  if (extra && distill_out_dev)   // x, distill_tokens = x[:, :-1], x[:, -1]  (distill.py:33)
    HIPCHK(hipMemcpy2DAsync(distill_out_dev, (size_t)d * 4, x_last + (int64_t)(ntok - 1) * d, (size_t)ntok * d * 4, (size_t)d * 4, b, hipMemcpyDeviceToDevice, e->stream));
  if ((rc = head_forward(e, x_last, b, head_tok, cait ? 0 : extra, logits_dev, err)) != VITX_OK) return rc;
  e->last_extra = extra;
  e->have_fwd = true;
  e->have_tf = false;
  e->last_b = b; e->last_np = np; e->last_ntok = ntok; e->last_H = H; e->last_W = W; e->last_training = training; e->last_seed = seed;
  return VITX_OK;
}

// ------------------------------------------------------------------------------------------------
// efficient.ViT (efficient.py:12-56): the model is a shell around a transformer object supplied by the caller --
// patch embedding + cls token + position embedding (efficient.py:40-46) in front of it, pooling + mlp_head (efficient.py:49-54)
// behind it.  The four pieces below are that shell and its VJP; the tokens cross the boundary as fp32 [b, n, dim].
// ------------------------------------------------------------------------------------------------
static int shell_check(const vitx_engine* e, std::string& err) {
  const vitx_config& c = e->cfg;
  if (c.variant != VITX_VARIANT_VIT && c.variant != VITX_VARIANT_DEEPVIT) { err = "embed / head entry points: ViT / DeepViT handles only"; return VITX_ERR_UNSUPPORTED; }
  if (c.num_parallel_branches > 1) { err = "embed / head entry points: not for parallel_vit handles"; return VITX_ERR_UNSUPPORTED; }
  return VITX_OK;
}

int engine_embed_forward(vitx_engine* e, const float* img_dev, int b, int H, int W, float* tokens_dev, std::string& err) {
  const vitx_config& c = e->cfg;
  int rc;
  if ((rc = shell_check(e, err)) != VITX_OK) return rc;
  if (b <= 0 || b > c.max_batch) { err = "batch must be in [1, max_batch]"; return VITX_ERR_INVALID; }
  if (H <= 0 || W <= 0 || H > c.image_h || W > c.image_w || H % c.patch_h || W % c.patch_w) {
    err = "image dimensions must be divisible by the patch size";            // efficient.py:18
    return VITX_ERR_INVALID;
  }
  const int np = (H / c.patch_h) * (W / c.patch_w), ntok = np + 1, d = c.dim, T = e->bf16;
  ensure_geometry(e, b, ntok);
  if (e->params_dirty) engine_refresh_weights(e);
  prepare_patch_rows(e, (int64_t)b * np);
  {
    Prof pr(e, "patch_unfold", 0, (double)b * H * W * c.channels * 4 + (double)b * np * e->pd_k * e->esz);
    launch_unfold(img_dev, e->patches, T, b, H, W, c.channels, c.patch_h, c.patch_w, e->pd_k, e->stream);   // efficient.py:23
  }
  {
    Prof pr(e, "cls_pos_row", 0, 0);
    launch_cls_pos_row(tokens_dev, e->params + e->cls, e->params + e->pos, b, ntok, d, d, e->stream);       // efficient.py:43-45, row 0
  }
  EpiParams ep;
  ep.out = tokens_dev; ep.ldo = d; ep.pos = e->params + e->pos; ep.ldr = d;
  ep.np = np; ep.ntok = ntok; ep.tok_off = 1;
  dense_fwd(e, e->patches, e->pd_k, b * np, e->patch, EPI_PATCH, ep);                                       // efficient.py:24 (+44-45)
  e->have_fwd = false;          // e->patches no longer describes a full forward
  e->have_embed = true;
  e->last_b = b; e->last_np = np; e->last_ntok = ntok; e->last_H = H; e->last_W = W; e->last_training = 0; e->last_extra = 0;
  return VITX_OK;
}

// patch_embedding.layers[1] on its own (mae.py:37, simmim.py:79, mpp.py:200): nn.Dense(units=dim) (vit.py:143) on rows of unfolded
// patches [rows, p1*p2*C] -> [rows, dim], no cls token, no position embedding.  Forward only.
int engine_patch_dense_forward(vitx_engine* e, const float* patches_dev, int rows, float* out_dev, std::string& err) {
  const vitx_config& c = e->cfg;
  if (rows <= 0 || (int64_t)rows > (int64_t)c.max_batch * e->np_max) { err = "patch_dense_forward: rows must be in [1, max_batch * num_patches]"; return VITX_ERR_INVALID; }
  if (e->params_dirty) engine_refresh_weights(e);
  prepare_patch_rows(e, rows);
  launch_convert(patches_dev, e->pd, e->patches, e->bf16, e->pd_k, rows, e->pd, e->pd_k, e->stream);
  EpiParams ep;
  ep.out = out_dev; ep.ldo = c.dim;
  dense_fwd(e, e->patches, e->pd_k, rows, e->patch, EPI_STORE_F32, ep);
  e->have_fwd = false; e->have_embed = false;   // e->patches no longer describes a forward that can be differentiated
  return VITX_OK;
}

int engine_head_forward(vitx_engine* e, const float* x_dev, int b, int n, float* logits_dev, std::string& err) {
  const vitx_config& c = e->cfg;
  int rc;
  if (b <= 0 || b > c.max_batch || n <= 0 || n > e->ntok_cap) { err = "head_forward: b or n out of range"; return VITX_ERR_INVALID; }
  e->have_fwd = false;          // the head state of a preceding full forward is overwritten: its backward must not run on it
  const int d = c.dim;
  if (e->params_dirty) engine_refresh_weights(e);
  if (!e->shell_x) DALLOC(e->shell_x, (size_t)e->mp * d * 4, false);
  if (x_dev != e->shell_x) HIPCHK(hipMemcpyAsync(e->shell_x, x_dev, (size_t)b * n * d * 4, hipMemcpyDeviceToDevice, e->stream));
  if (e->bf16) {   // rows [b, next multiple of 64) of the normalised input are K padding of the head's weight-gradient GEMM
    const int64_t r1 = round_up(b, 64);
    if (r1 > b) HIPCHK(hipMemsetAsync(boff(e->yh, (int64_t)b * d, 2), 0, (size_t)(r1 - b) * d * 2, e->stream));
  }
  if ((rc = head_forward(e, e->shell_x, b, n, 0, logits_dev, err)) != VITX_OK) return rc;                   // efficient.py:49-54
  e->have_head = true; e->shell_b = b; e->shell_n = n;
  return VITX_OK;
}

// d(logits) [b, num_classes] (NULL: the engine's internal dlogits, e.g. from vitx_ce_loss_grad_dev) -> d(x) [b, n, dim];
// fills the mlp_head.* entries of the gradient arena
int engine_head_backward(vitx_engine* e, const float* dlogits_dev, float* dx_dev, std::string& err) {
  const vitx_config& c = e->cfg;
  if (!e->have_head) { err = "head_backward requires a preceding head_forward"; return VITX_ERR_STATE; }
  const int b = e->shell_b, n = e->shell_n, d = c.dim, T = e->bf16, nc = c.num_classes;
  if (dlogits_dev)
    HIPCHK(hipMemcpy2DAsync(e->dlogits, (size_t)e->nc_k * 4, dlogits_dev, (size_t)nc * 4, (size_t)nc * 4, b, hipMemcpyDeviceToDevice, e->stream));
  {
    Prof pr(e, "head_prep", 0, 0);
    launch_colsum(e->dlogits, 0, e->nc_k, b, nc, e->red_ws, e->grads + e->head.b, e->stream);
    if (T) {
      const int64_t r1 = round_up(b, 64);
      if (r1 > b) HIPCHK(hipMemsetAsync(boff(e->dl_lp, (int64_t)b * e->nc_k, 2), 0, (size_t)(r1 - b) * e->nc_k * 2, e->stream));
      launch_convert(e->dlogits, e->nc_k, e->dl_lp, 1, e->nc_k, b, nc, e->nc_k, e->stream);
    }
  }
  const void* dlT = T ? e->dl_lp : (const void*)e->dlogits;
  {
    EpiParams ep; ep.out = e->dyh; ep.ldo = d;
    dense_dgrad(e, dlT, e->nc_k, b, e->head, EPI_STORE, ep);
  }
  dense_wgrad(e, e->yh, d, dlT, e->nc_k, b, e->head);
  const int rows = b * n;
  { Prof pr(e, "fill_zero", 0, (double)rows * d * 4); launch_fill_zero(e->g, (int64_t)round_up(rows, 256) * d * 4, e->stream); }
  {
    Prof pr(e, "layernorm_bwd", 0, 0);
    if (c.pool == VITX_POOL_MEAN) {
      launch_layernorm_bwd(e->dyh, T, d, e->pooled, d, e->mean_h, e->rstd_h, e->params + e->head_g, nullptr, 0, e->dpooled, d, nullptr, 0,
                           e->red_ws, e->grads + e->head_g, e->grads + e->head_b, nullptr, b, d, e->stream);
      launch_mean_pool_bwd(e->dpooled, b, n, d, e->g, e->stream, n);
    } else {
      const int64_t ldrow = (int64_t)n * d;
      launch_layernorm_bwd(e->dyh, T, d, e->shell_x, ldrow, e->mean_h, e->rstd_h, e->params + e->head_g, nullptr, 0, e->g, ldrow, nullptr, 0,
                           e->red_ws, e->grads + e->head_g, e->grads + e->head_b, nullptr, b, d, e->stream);
    }
  }
  if (dx_dev && dx_dev != e->g) HIPCHK(hipMemcpyAsync(dx_dev, e->g, (size_t)rows * d * 4, hipMemcpyDeviceToDevice, e->stream));
  return VITX_OK;
}

// d(tokens) [b, np + 1, dim] -> the pos_embedding, cls_token and patch_embedding.* entries of the gradient arena (+ optional d(img))
int engine_embed_backward(vitx_engine* e, const float* dtokens_dev, float* dimg_dev, std::string& err) {
  const vitx_config& c = e->cfg;
  if (!e->have_embed) { err = "embed_backward requires a preceding embed_forward"; return VITX_ERR_STATE; }
  const int b = e->last_b, np = e->last_np, ntok = e->last_ntok, d = c.dim, T = e->bf16;
  if (dtokens_dev != e->g) HIPCHK(hipMemcpyAsync(e->g, dtokens_dev, (size_t)b * ntok * d * 4, hipMemcpyDeviceToDevice, e->stream));
  {
    Prof pr(e, "embed_bwd", 0, (double)b * ntok * d * 4);
    launch_extract_rows(e->g, b, ntok, 1, np, d, e->d_y, T, d, e->stream);                    // dE = g[:, 1:, :]
  }
  // the projection's weight gradient first: on the side stream it runs beside the batch sums below (which only read g) instead of behind them
  dense_wgrad(e, e->patches, e->pd_k, e->d_y, d, b * np, e->patch);
  {
    Prof pr(e, "embed_bwd", 0, (double)b * ntok * d * 4);
    if (ntok < e->ntok_max)   // position rows the image did not reach (efficient.py:45 slices the table)
      launch_fill_zero(e->grads + e->pos + (int64_t)ntok * d, (int64_t)(e->ntok_max - ntok) * d * 4, e->stream);
    launch_batch_reduce(e->g, b, ntok, d, 0, ntok, e->grads + e->pos, e->stream);             // dpos[j] = sum_b g[b,j]
    launch_batch_reduce(e->g, b, ntok, d, 0, 1, e->grads + e->cls, e->stream);                // dcls = sum_b g[b,0]
    launch_sum_rows(e->grads + e->pos + d, np, d, e->grads + e->patch.b, e->stream);          // db = sum over patch rows
  }
  if (dimg_dev) {
    EpiParams ep; ep.out = e->tmp_f32; ep.ldo = e->pd;
    dense_dgrad(e, e->d_y, d, b * np, e->patch, EPI_STORE_F32, ep);
    Prof pr(e, "patch_fold", 0, 0);
    launch_fold_add(e->tmp_f32, e->pd, dimg_dev, b, e->last_H, e->last_W, c.channels, c.patch_h, c.patch_w, e->stream);
  }
  return VITX_OK;
}

// training != 0 with a dropout rate > 0 draws the Dropout masks of vit.py:41,43,64 from (seed, site, element) exactly as the full
// forward does (the reference calls self.encoder.transformer(tokens, training=training), mae.py:69, simmim.py:116, efficient.py:47);
// the backward below regenerates the same masks.
int engine_transformer_forward(vitx_engine* e, const float* tokens_dev, int b, int n, int training, uint64_t seed, float* out_dev, std::string& err) {
  const vitx_config& c = e->cfg;
  if (c.variant == VITX_VARIANT_CAIT || c.variant == VITX_VARIANT_PATCH_MERGER) { err = "transformer_forward: ViT / DeepViT only"; return VITX_ERR_UNSUPPORTED; }
  if (b <= 0 || b > c.max_batch || n <= 0 || n > e->ntok_cap) { err = "transformer_forward: b or n out of range"; return VITX_ERR_INVALID; }
  ensure_geometry(e, b, n);
  if (e->params_dirty) engine_refresh_weights(e);
  Stage& s0 = e->stages[0];
  const size_t bytes = (size_t)b * n * c.dim * 4;
  if (s0.depth == 0) { HIPCHK(hipMemcpyAsync(out_dev, tokens_dev, bytes, hipMemcpyDeviceToDevice, e->stream)); return VITX_OK; }
  HIPCHK(hipMemcpyAsync(s0.ba[0].x_in, tokens_dev, bytes, hipMemcpyDeviceToDevice, e->stream));
  int rc;
  const float drop = training ? c.dropout : 0.f;
  for (int l = 0; l < s0.depth; ++l)
    if ((rc = block_forward(e, s0, 0, l, b, n, 0, nullptr, drop, seed, err)) != VITX_OK) return rc;
  HIPCHK(hipMemcpyAsync(out_dev, s0.ba[s0.depth - 1].x_out, bytes, hipMemcpyDeviceToDevice, e->stream));
  e->have_fwd = false;   // saved activations no longer describe a full model forward
  e->have_tf = true; e->tf_b = b; e->tf_n = n; e->tf_drop = drop; e->tf_seed = seed;
  e->last_training = 0;
  return VITX_OK;
}

// VJP of engine_transformer_forward: d(out) [b,n,dim] -> d(tokens) [b,n,dim] and the gradients of the transformer's parameters
// (every other entry of the gradient arena is zero).  This is what lets the wrappers that call encoder.transformer(tokens)
// on a subset of the patches (mae.py:69, simmim.py:116) train the encoder.
int engine_transformer_backward(vitx_engine* e, const float* dout_dev, float* dtokens_dev, std::string& err) {
  const vitx_config& c = e->cfg;
  if (!e->have_tf) { err = "transformer_backward requires a preceding transformer_forward"; return VITX_ERR_STATE; }
  const int b = e->tf_b, n = e->tf_n, d = c.dim, T = e->bf16;
  Stage& s0 = e->stages[0];
  const size_t bytes = (size_t)b * n * d * 4;
  { Prof pr(e, "fill_zero", 0, (double)e->n_arena * 4); launch_fill_zero(e->grads, (int64_t)e->n_arena * 4, e->stream); }
  launch_fill_zero(e->g, (int64_t)round_up((int64_t)b * n, 256) * d * 4, e->stream);
  HIPCHK(hipMemcpyAsync(e->g, dout_dev, bytes, hipMemcpyDeviceToDevice, e->stream));
  if (T) launch_convert(e->g, d, e->g_lp, 1, d, b * n, d, d, e->stream);
  int rc;
  SideScope side_scope(e, true);
  e->dbr_ready = 0;
  for (int l = s0.depth - 1; l >= 0; --l)
    if ((rc = block_backward(e, s0, 0, l, b, n, 0, e->tf_drop, e->tf_seed, err, l - 1)) != VITX_OK) return rc;
  if (dtokens_dev) HIPCHK(hipMemcpyAsync(dtokens_dev, e->g, bytes, hipMemcpyDeviceToDevice, e->stream));
  return VITX_OK;
}

// The first two layers of encoder.patch_embedding plus the position rows the wrappers add themselves (mae.py:49-55,
// simmim.py:88-100): tokens[b, p] = patches[b, p] @ W + bias + pos_embedding[0, 1 + p], no cls row.  Optionally also the fp32
// patches (the reconstruction target, mae.py:65 / simmim.py:125).
int engine_patch_tokens_forward(vitx_engine* e, const float* img_dev, int b, int H, int W, float* tokens_dev, float* patches_f32_dev,
                                std::string& err) {
  const vitx_config& c = e->cfg;
  if (c.variant == VITX_VARIANT_CAIT || c.variant == VITX_VARIANT_PATCH_MERGER) { err = "patch_tokens_forward: ViT / DeepViT only"; return VITX_ERR_UNSUPPORTED; }
  if (b <= 0 || b > c.max_batch) { err = "batch must be in [1, max_batch]"; return VITX_ERR_INVALID; }
  if (H <= 0 || W <= 0 || H > c.image_h || W > c.image_w || H % c.patch_h || W % c.patch_w) {
    err = "Image dimensions must be divisible by the patch size.";
    return VITX_ERR_INVALID;
  }
  const int np = (H / c.patch_h) * (W / c.patch_w), d = c.dim, T = e->bf16;
  if (e->params_dirty) engine_refresh_weights(e);
  prepare_patch_rows(e, (int64_t)b * np);
  {
    Prof pr(e, "patch_unfold", 0, (double)b * H * W * c.channels * 4 + (double)b * np * e->pd_k * e->esz);
    launch_unfold(img_dev, e->patches, T, b, H, W, c.channels, c.patch_h, c.patch_w, e->pd_k, e->stream);
    if (patches_f32_dev) launch_unfold(img_dev, patches_f32_dev, 0, b, H, W, c.channels, c.patch_h, c.patch_w, e->pd, e->stream);
  }
  EpiParams ep;
  ep.out = tokens_dev; ep.ldo = d; ep.pos = e->params + e->pos + d; ep.ldr = d;   // pos row of patch p is 1 + p
  ep.np = np; ep.ntok = np; ep.tok_off = 0;
  dense_fwd(e, e->patches, e->pd_k, b * np, e->patch, EPI_PATCH, ep);
  e->have_pt = true; e->pt_b = b; e->pt_np = np;
  return VITX_OK;
}

// VJP of the call above: d(tokens) [b, np, dim] -> gradient-arena entries of patch_embedding.kernel / .bias and of
// pos_embedding rows 1..np (overwritten; call it after transformer_backward, which clears the arena).
int engine_patch_tokens_backward(vitx_engine* e, const float* dtokens_dev, std::string& err) {
  if (!e->have_pt) { err = "patch_tokens_backward requires a preceding patch_tokens_forward"; return VITX_ERR_STATE; }
  const int b = e->pt_b, np = e->pt_np, d = e->cfg.dim, T = e->bf16;
  const int64_t rows = (int64_t)b * np;
  if (!e->pt_dy) DALLOC(e->pt_dy, (size_t)e->mpp * d * e->esz, false);
  if (e->pt_dy_rows >= 0 && e->pt_dy_rows != rows && T) HIPCHK(hipMemsetAsync(e->pt_dy, 0, (size_t)e->mpp * d * e->esz, e->stream));
  e->pt_dy_rows = rows;
  {
    Prof pr(e, "embed_bwd", 0, (double)rows * d * 4);
    float* dpos = e->grads + e->pos + d;
    launch_batch_reduce(dtokens_dev, b, np, d, 0, np, dpos, e->stream);                  // dpos[1 + p] = sum_b dtokens[b, p]
    launch_sum_rows(dpos, np, d, e->grads + e->patch.b, e->stream);                      // dbias = sum over every patch row
    launch_extract_rows(dtokens_dev, b, np, 0, np, d, e->pt_dy, T, d, e->stream);        // T copy (identity in parity mode)
  }
  dense_wgrad(e, e->patches, e->pd_k, e->pt_dy, d, (int)rows, e->patch);
  return VITX_OK;
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
int engine_backward(vitx_engine* e, const float* dlogits_dev, float* dimg_dev, std::string& err, const float* d_distill_dev, float* d_token_out_dev) {
  const vitx_config& c = e->cfg;
  if (!e->have_fwd) { err = "backward requires a preceding forward"; return VITX_ERR_STATE; }
  const int extra = e->last_extra;
  const bool cait = c.variant == VITX_VARIANT_CAIT, merger = c.variant == VITX_VARIANT_PATCH_MERGER;
  const bool no_cls = cait || merger;
  const bool merging = merger && e->merge_after >= 0;
  const int b = e->last_b, np = e->last_np, ntok = e->last_ntok, d = c.dim, T = e->bf16;
  const int nc = c.num_classes;
  const float drop = e->last_training ? c.dropout : 0.f;
  // weight gradients of this pass on the side stream (not with the PatchMerger: its row-extent change re-zeroes K padding in place)
  SideScope side_scope(e, !merging);
  {
    // every gradient tensor is overwritten by its producer; a full clear is only needed when part of the arena will not be
    // produced this step (pos_embedding rows beyond the image's tokens, blocks skipped by CaiT layer dropout)
    bool need_clear = ntok - extra < e->ntok_max;
    for (auto& k : e->layer_kept) for (bool kept : k) need_clear = need_clear || !kept;
    if (need_clear) { Prof pr(e, "fill_zero", 0, (double)e->n_arena * 4); launch_fill_zero(e->grads, (int64_t)e->n_arena * 4, e->stream); }
  }
  if (dlogits_dev)
    HIPCHK(hipMemcpy2DAsync(e->dlogits, (size_t)e->nc_k * 4, dlogits_dev, (size_t)nc * 4, (size_t)nc * 4, b, hipMemcpyDeviceToDevice, e->stream));

  // ---- head: logits = LN(pooled) @ W + b   (vit.py:154-157)
  Stage& s_last = e->stages.back();
  Stage& s0 = e->stages[0];
  const float* x_last;
  int head_tok;
  if (cait) {
    Stage& s1 = e->stages[1];
    x_last = s1.depth > 0 ? s1.ba[s1.depth - 1].x_out : e->pooled;
    head_tok = 1;
  } else {
    x_last = s0.depth > 0 ? s0.ba[s0.depth - 1].x_out : e->tmp_f32;
    head_tok = ntok;
    if (merging) {
      head_tok = e->merge_t;
      if (e->merge_after == s0.depth - 1) x_last = e->pm_out;
      // The shared bf16 gradient buffers were last used at b*ntok rows and are about to be used at b*merge_t: the rows between the
      // new extent and its next multiple of 64 are K padding of the weight-gradient GEMMs and must read as zero.
      if (T) {
        const int64_t r0 = (int64_t)b * e->merge_t, r1 = round_up(r0, 64);
        if (r1 > r0) {
          HIPCHK(hipMemsetAsync(boff(e->g_lp, r0 * d, 2), 0, (size_t)(r1 - r0) * d * 2, e->stream));
          HIPCHK(hipMemsetAsync(boff(e->d_br, r0 * d, 2), 0, (size_t)(r1 - r0) * d * 2, e->stream));
          HIPCHK(hipMemsetAsync(boff(e->d_qkv, r0 * 3 * e->inner, 2), 0, (size_t)(r1 - r0) * 3 * e->inner * 2, e->stream));
          HIPCHK(hipMemsetAsync(boff(e->d_o, r0 * e->inner, 2), 0, (size_t)(r1 - r0) * e->inner * 2, e->stream));
        }
      }
    }
  }
  (void)s_last;
  {
    Prof pr(e, "head_prep", 0, 0);
    launch_colsum(e->dlogits, 0, e->nc_k, b, nc, e->red_ws, e->grads + e->head.b, e->stream);
    if (T) launch_convert(e->dlogits, e->nc_k, e->dl_lp, 1, e->nc_k, b, nc, e->nc_k, e->stream);
  }
  const void* dlT = T ? e->dl_lp : (const void*)e->dlogits;
  {
    EpiParams ep; ep.out = e->dyh; ep.ldo = d;
    dense_dgrad(e, dlT, e->nc_k, b, e->head, EPI_STORE, ep);
  }
  dense_wgrad(e, e->yh, d, dlT, e->nc_k, b, e->head);
  const bool mean_pool = !cait && c.pool == VITX_POOL_MEAN;
  const int head_rows = b * head_tok;
  { Prof pr(e, "fill_zero", 0, (double)head_rows * d * 4); launch_fill_zero(e->g, (int64_t)round_up(head_rows, 256) * d * 4, e->stream); }
  {
    Prof pr(e, "layernorm_bwd", 0, 0);
    if (mean_pool) {
      launch_layernorm_bwd(e->dyh, T, d, e->pooled, d, e->mean_h, e->rstd_h, e->params + e->head_g, nullptr, 0, e->dpooled, d, nullptr, 0,
                           e->red_ws, e->grads + e->head_g, e->grads + e->head_b, nullptr, b, d, e->stream);
      launch_mean_pool_bwd(e->dpooled, b, head_tok - extra, d, e->g, e->stream, head_tok);
    } else {
      const int64_t ldrow = (int64_t)head_tok * d;
      launch_layernorm_bwd(e->dyh, T, d, x_last, ldrow, e->mean_h, e->rstd_h, e->params + e->head_g, nullptr, 0, e->g, ldrow, nullptr, 0,
                           e->red_ws, e->grads + e->head_g, e->grads + e->head_b, nullptr, b, d, e->stream);
    }
    if (extra && d_distill_dev)   // cotangent of the split-off distillation token: the last row of every image
      HIPCHK(hipMemcpy2DAsync(e->g + (int64_t)(ntok - 1) * d, (size_t)ntok * d * 4, d_distill_dev, (size_t)d * 4, (size_t)d * 4, b, hipMemcpyDeviceToDevice, e->stream));
    if (T) launch_convert(e->g, d, e->g_lp, 1, d, head_rows, d, d, e->stream);
  }
  report_ready(e, e->head_g, e->n_arena - e->head_g);

  int rc;
  e->dbr_ready = 0;
  auto next_kept = [&](int si, int l) {   // the next lower layer of the stage that survived layer dropout in this forward (cait.py:17-31), or -1
    for (int j = l - 1; j >= 0; --j) if (e->layer_kept[(size_t)si][(size_t)j]) return j;
    return -1;
  };
  if (cait) {
    Stage& s1 = e->stages[1];
    launch_fill_zero(e->g_ctx, (int64_t)b * np * d * 4, e->stream);
    for (int l = s1.depth - 1; l >= 0; --l) {
      if (e->layer_kept[1][(size_t)l]) {
        if ((rc = block_backward(e, s1, 1, l, b, 1, np, drop, e->last_seed, err, next_kept(1, l))) != VITX_OK) return rc;
      } else {
#ifndef VITX_COMM_LEGACY_ORDER   // (defined only in the variant build that shows the two-rank test catching the round-5 behaviour, comm.hip)
        // a dropped layer's gradients are final zeros (every rank reports every arena range, whatever its layer-dropout draw)
        report_ready(e, s1.bp[(size_t)l].p_begin, s1.bp[(size_t)l].p_end - s1.bp[(size_t)l].p_begin);
#endif
      }
    }
    {
      Prof pr(e, "embed_bwd", 0, 0);
      launch_batch_reduce(e->g, b, 1, d, 0, 1, e->grads + e->cls, e->stream);          // dcls = sum_b g (cait.py:189 VJP)
      // the patch stage's output gradient is what flowed back through every cls layer's context
      HIPCHK(hipMemcpyAsync(e->g, e->g_ctx, (size_t)b * np * d * 4, hipMemcpyDeviceToDevice, e->stream));
      if (T) launch_convert(e->g, d, e->g_lp, 1, d, b * np, d, d, e->stream);
    }
  }
  for (int l = s0.depth - 1; l >= 0; --l) {
    if (merging && l == e->merge_after && (rc = merger_backward(e, s0.ba[l].x_out, b, ntok, err)) != VITX_OK) return rc;
    const int rows_tok = (merging && l > e->merge_after) ? e->merge_t : ntok;
    if (e->layer_kept[0][(size_t)l]) {
      if ((rc = block_backward(e, s0, 0, l, b, rows_tok, 0, drop, e->last_seed, err, merging ? -1 : next_kept(0, l))) != VITX_OK) return rc;
    } else {
#ifndef VITX_COMM_LEGACY_ORDER
      report_ready(e, s0.bp[(size_t)l].p_begin, s0.bp[(size_t)l].p_end - s0.bp[(size_t)l].p_begin);   // dropped layer: final zeros
#endif
    }
  }
  if (e->last_training && c.emb_dropout > 0.f) {
    Prof pr(e, "dropout", 0, 0);
    launch_dropout(e->g, 0, (int64_t)b * ntok * d, c.emb_dropout, e->last_seed, 0u, e->stream);   // same mask as the forward (vit.py:166)
  }

  // ---- embedding: x0 = [cls | patches @ W + b] + pos   (vit.py:160-165; cait.py:181-184)
  const int tok_off = no_cls ? 0 : 1;
  {
    Prof pr(e, "embed_bwd", 0, 0);
    launch_extract_rows(e->g, b, ntok, tok_off, np, d, e->d_y, T, d, e->stream);        // dE = g[:, tok_off:, :]
  }
  // the projection's weight gradient first: on the side stream it runs beside the batch sums below (which only read g) instead of behind them
  dense_wgrad(e, e->patches, e->pd_k, e->d_y, d, b * np, e->patch);
  {
    Prof pr(e, "embed_bwd", 0, (double)b * ntok * d * 4);
    launch_batch_reduce(e->g, b, ntok, d, 0, ntok - extra, e->grads + e->pos, e->stream);      // dpos[j] = sum_b g[b,j]
    if (extra && d_token_out_dev) launch_batch_reduce(e->g, b, ntok, d, ntok - 1, 1, d_token_out_dev, e->stream);   // d(distill token) = sum_b g[b,-1]
    if (!no_cls) launch_batch_reduce(e->g, b, ntok, d, 0, 1, e->grads + e->cls, e->stream);   // dcls = sum_b g[b,0]
    launch_sum_rows(e->grads + e->pos + (int64_t)tok_off * d, np, d, e->grads + e->patch.b, e->stream);   // db = sum over patch rows
  }
  if (dimg_dev) {
    EpiParams ep; ep.out = e->tmp_f32; ep.ldo = e->pd;
    dense_dgrad(e, e->d_y, d, b * np, e->patch, EPI_STORE_F32, ep);
    if (e->last_from_patches) {   // the forward took patch rows: d(patches) [b, np, pd] goes out as it is
      HIPCHK(hipMemcpyAsync(dimg_dev, e->tmp_f32, (size_t)b * np * e->pd * 4, hipMemcpyDeviceToDevice, e->stream));
    } else {
      Prof pr(e, "patch_fold", 0, 0);
      launch_fold_add(e->tmp_f32, e->pd, dimg_dev, b, e->last_H, e->last_W, c.channels, c.patch_h, c.patch_w, e->stream);
    }
  }
  report_ready(e, 0, e->stages[0].depth > 0 ? e->stages[0].bp[0].p_begin : e->head_g);
  return VITX_OK;
}

// ------------------------------------------------------------------------------------------------
// raw GEMM micro-benchmark (random bf16 operands), checked against the generic fp32-FMA kernel
// ------------------------------------------------------------------------------------------------
// Full-size self-check of the bf16 MFMA GEMMs, everything on the device (the launches of the benchmarked step have 50k rows: too large for the
// host-side comparison of engine_bench_gemm).  kind 0: the NT kernel `kernel` with fused epilogue `epilogue` (codes of engine_bench_gemm) against the
// k-ordered fp32-FMA kernel running the SAME epilogue functor on the same bf16 operands; kind 1: the weight-gradient (TN) kernel with the engine's
// split-K rule + the fixed-order slice reduction against the fp32-FMA kernel (M = in, N = out, K = token rows).
// err[0] = max |got - want| / (1 + |want|) over every output element, err[1] = the same for the second output / the fused column sums (or -1).
int engine_check_gemm(vitx_engine* e, int kind, int M, int N, int K, int kernel, int epilogue, float* errs, std::string& err) {
  if (K % 64 || M <= 0 || N <= 0 || kind < 0 || kind > 1 || epilogue < 0 || epilogue > 4) { err = "check_gemm: bad arguments (K must be a multiple of 64)"; return VITX_ERR_INVALID; }
  float* dmax;
  HIPCHK(hipMalloc((void**)&dmax, 16));
  HIPCHK(hipMemsetAsync(dmax, 0, 16, e->stream));
  errs[0] = errs[1] = -1.f;
  std::vector<void*> bufs;
  auto alloc = [&](void** p, size_t bytes) -> hipError_t { hipError_t r = hipMalloc(p, bytes + 8192); if (r == hipSuccess) { bufs.push_back(*p); r = hipMemsetAsync(*p, 0, bytes + 8192, e->stream); } return r; };
  auto cleanup = [&]() { for (void* b : bufs) (void)hipFree(b); (void)hipFree(dmax); };
  if (kind == 0) {
    const int64_t Mp = round_up(M, 1280) + 320, Np = round_up(N, 256);   // (+ 320: a tail launch's 192-row tiles start at a row that is no multiple of 192)
    bf16_t *A, *B, *T1[2], *T2[2], *aux; float *F[2], *bias, *R, *cs[2];
    if (alloc((void**)&A, (size_t)Mp * K * 2) || alloc((void**)&B, (size_t)Np * K * 2) || alloc((void**)&bias, (size_t)Np * 4) || alloc((void**)&R, (size_t)Mp * Np * 4) ||
        alloc((void**)&aux, (size_t)Mp * Np * 2) || alloc((void**)&cs[0], (size_t)(Mp / 96 + 8) * Np * 4) || alloc((void**)&cs[1], (size_t)Np * 4 * 64)) { cleanup(); err = "check_gemm: out of memory"; return VITX_ERR_HIP; }
    for (int i = 0; i < 2; ++i)
      if (alloc((void**)&T1[i], (size_t)Mp * Np * 2) || alloc((void**)&T2[i], (size_t)Mp * Np * 2) || alloc((void**)&F[i], (size_t)Mp * Np * 4)) { cleanup(); err = "check_gemm: out of memory"; return VITX_ERR_HIP; }
    launch_fill_random_bf16(A, (int64_t)M * K, 11u, 1.0f, e->stream);          // rows >= M stay zero (row padding invariant)
    launch_fill_random_bf16(B, (int64_t)N * K, 12u, 1.0f / 16.f, e->stream);   // |acc| ~ sqrt(K) / 16: pre-activations of a sensible size for the GELU forms
    launch_fill_random_bf16(aux, (int64_t)M * Np, 13u, 1.0f, e->stream);
    {   // bias, residual: random bf16 patterns widened to fp32
      bf16_t* t;
      if (alloc((void**)&t, (size_t)Mp * Np * 2)) { cleanup(); err = "check_gemm: out of memory"; return VITX_ERR_HIP; }
      launch_fill_random_bf16(t, (int64_t)M * Np, 14u, 2.0f, e->stream);
      launch_to_f32(t, 1, Np, R, Np, M, (int)Np, e->stream);
      launch_fill_random_bf16(t, Np, 15u, 0.5f, e->stream);
      launch_to_f32(t, 1, Np, bias, Np, 1, (int)Np, e->stream);
    }
    Bf16GemmArgs g;
    g.A = A; g.lda = K; g.B = B; g.ldb = K; g.M = M; g.N = N; g.K = K; g.kernel = kernel & (15 | 256 | 512);
    g.tail = (kernel >> 10) & 15;   // (round 6) bits 10..13: tile variant of the tail launch (tail balancing, gemm_bf16.hip)
    GenericGemmArgs gg;
    gg.A = A; gg.B = B; gg.M = M; gg.N = N; gg.K = K; gg.sam = K; gg.sak = 1; gg.sbk = 1; gg.sbn = K;
    int mode = EPI_STORE_F32;
    auto params = [&](int i) {
      EpiParams ep;
      ep.M = M; ep.N = N; ep.zero_pad = 1;
      if (epilogue == 1) { mode = EPI_BIAS_RESID; ep.out = F[i]; ep.ldo = Np; ep.resid = R; ep.ldr = Np; ep.bias = bias; }
      else if (epilogue == 2) { mode = EPI_BIAS_GELU; ep.out = T1[i]; ep.ldo = Np; ep.out2 = T2[i]; ep.ldo2 = Np; ep.bias = bias; }
      else if (epilogue == 3) { mode = EPI_STORE; ep.out = T1[i]; ep.ldo = Np; }
      else if (epilogue == 4) { mode = EPI_GELU_BWD; ep.out = T1[i]; ep.ldo = Np; ep.aux = aux; ep.ldaux = Np; if (i == 0) { ep.colsum = cs[0]; ep.ldcs = Np; } }
      else { ep.out = F[i]; ep.ldo = Np; }
      finalize_epi(ep);
      return ep;
    };
    const EpiParams ep0 = params(0), ep1 = params(1);
    launch_gemm_bf16(g, ep0, mode, e->stream);
    launch_gemm_generic(gg, ep1, mode, 1, 1, 1, e->stream);
    if (epilogue == 0 || epilogue == 1) launch_max_rel_diff(F[0], F[1], 0, M, N, Np, Np, dmax, e->stream);
    else launch_max_rel_diff(T1[0], T1[1], 1, M, N, Np, Np, dmax, e->stream);
    if (epilogue == 2) launch_max_rel_diff(T2[0], T2[1], 1, M, N, Np, Np, dmax + 1, e->stream);
    if (epilogue == 4) {   // fused column sums (one partial row per M-tile -- or per wave row of a tile -- of the launch; unwritten rows are zero)
      const int nt = (int)ceil_div(M, 96);   // >= the wave rows of every variant (rows no variant writes are zero)
      float* ws;
      if (alloc((void**)&ws, (size_t)std::max<int64_t>(colsum_ws_elems(N), (int64_t)nt * Np) * 4)) { cleanup(); err = "check_gemm: out of memory"; return VITX_ERR_HIP; }
      if (g.kernel == 0) { cleanup(); err = "check_gemm: epilogue 4 needs an explicit kernel variant (the column-sum rows follow its tile height)"; return VITX_ERR_INVALID; }
      launch_reduce_partials(cs[0], nt, Np, N, cs[1], 1.0f, e->stream);
      launch_colsum(T1[1], 1, Np, M, N, ws, cs[1] + Np, e->stream);
      launch_max_rel_diff(cs[1], cs[1] + Np, 0, 1, N, Np, Np, dmax + 1, e->stream);
    }
  } else {
    const int in = M, out = N, tokens = K;
    bf16_t *X, *dY; float *dW[2], *part;
    const int tile = gemm_bf16_tn_tile(kernel, in, out);
    const int64_t tiles = ceil_div(in, tile) * ceil_div(out, tile);
    const int nk = tokens / 64;
    const int split = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(nk, 4), std::max<int64_t>(1, 256 / tiles)));   // dense_wgrad's rule
    const int slices = gemm_bf16_num_slices(tokens, split);
    const int64_t ldx = round_up(in, 256), ldy = round_up(out, 256);
    if (alloc((void**)&X, (size_t)(tokens + 64) * ldx * 2) || alloc((void**)&dY, (size_t)(tokens + 64) * ldy * 2) || alloc((void**)&dW[0], (size_t)in * out * 4) ||
        alloc((void**)&dW[1], (size_t)in * out * 4) || alloc((void**)&part, (size_t)slices * in * out * 4)) { cleanup(); err = "check_gemm: out of memory"; return VITX_ERR_HIP; }
    launch_fill_random_bf16(X, (int64_t)tokens * ldx, 21u, 1.0f, e->stream);
    launch_fill_random_bf16(dY, (int64_t)tokens * ldy, 22u, 1.0f / 16.f, e->stream);
    Bf16GemmArgs g;
    g.A = X; g.lda = ldx; g.B = dY; g.ldb = ldy; g.M = in; g.N = out; g.K = tokens; g.kernel = kernel; g.split_k = split;
    EpiParams ep;
    ep.out = slices == 1 ? dW[0] : part; ep.ldo = out; ep.partial_stride = (int64_t)in * out; ep.M = in; ep.N = out;
    finalize_epi(ep);
    launch_gemm_bf16_tn(g, ep, e->stream);
    if (slices > 1) launch_reduce_partials(part, slices, (int64_t)in * out, (int64_t)in * out, dW[0], 1.0f, e->stream);
    GenericGemmArgs gg;
    gg.A = X; gg.B = dY; gg.M = in; gg.N = out; gg.K = tokens; gg.sam = 1; gg.sak = ldx; gg.sbk = ldy; gg.sbn = 1;
    EpiParams e2; e2.out = dW[1]; e2.ldo = out; e2.M = in; e2.N = out;
    finalize_epi(e2);
    launch_gemm_generic(gg, e2, EPI_STORE_F32, 1, 1, 0, e->stream);
    launch_max_rel_diff(dW[0], dW[1], 0, in, out, out, out, dmax, e->stream);
    errs[1] = (float)slices;
  }
  float h[2] = {0.f, 0.f};
  hipError_t rc = hipMemcpyAsync(h, dmax, 8, hipMemcpyDeviceToHost, e->stream);
  if (rc == hipSuccess) rc = hipStreamSynchronize(e->stream);
  cleanup();
  if (rc != hipSuccess) { err = std::string("check_gemm: ") + hipGetErrorString(rc); return VITX_ERR_HIP; }
  errs[0] = h[0];
  if (kind == 0 && (epilogue == 2 || epilogue == 4)) errs[1] = h[1];
  return VITX_OK;
}

int engine_bench_gemm(vitx_engine* e, int M, int N, int K, int kernel, int epilogue, int iters, float* avg_ms, float* max_err,
                      std::string& err) {
  if (K % 64 || M <= 0 || N <= 0) { err = "bench_gemm: K must be a multiple of 64"; return VITX_ERR_INVALID; }
  // bits 4..7 of `kernel` reach the kernels' `stagger` field, whose low bits double as timing-experiment switches (no DMA wait /
  // no DMA issue: WRONG results, faster launches).  A sweep that packs anything else into those bits measures the switch, not its
  // own parameter (profiles/r2/gemm_tile_band_README.txt), so they are refused unless the caller says it wants the experiment.
  if (((kernel >> 4) & 7) && !vitx_env("VITX_GEMM_XP")) { err = "vitx_bench_gemm: kernel bits 4-6 are timing-experiment switches (results invalid); set VITX_GEMM_XP=1 to use them"; return VITX_ERR_INVALID; }
  const int64_t Mp = round_up(M, 1280) + 320, Np = round_up(N, 256);   // (+ 320: a tail launch's 192-row tiles start at a row that is no multiple of 192)
  bf16_t *A, *B; float *C, *R, *bias; bf16_t* C2;
  HIPCHK(hipMalloc((void**)&A, (size_t)Mp * K * 2));
  HIPCHK(hipMalloc((void**)&B, (size_t)Np * K * 2));
  HIPCHK(hipMalloc((void**)&C, (size_t)Mp * Np * 4));
  HIPCHK(hipMalloc((void**)&R, (size_t)Mp * Np * 4));
  HIPCHK(hipMalloc((void**)&C2, (size_t)Mp * Np * 2 * 2));
  HIPCHK(hipMalloc((void**)&bias, (size_t)Np * 4));
  const float fill_scale = vitx_env("VITX_BENCH_ZERO") ? 0.0f : 1.0f;   // zero operands: DVFS / power-limit experiment only
  launch_fill_random_bf16(A, Mp * K, 1u, fill_scale, e->stream);
  launch_fill_random_bf16(B, Np * K, 2u, fill_scale, e->stream);
  // small problems are checked against the generic kernel: give the fused epilogues non-trivial bias / residual operands
  const bool check = max_err && (int64_t)M * N <= (1 << 25);
  std::vector<float> hbias((size_t)Np, 0.f), hres;
  HIPCHK(hipMemsetAsync(R, 0, (size_t)Mp * Np * 4, e->stream));
  HIPCHK(hipMemsetAsync(bias, 0, (size_t)Np * 4, e->stream));
  if (check && epilogue != 0) {
    for (int64_t j = 0; j < Np; ++j) hbias[j] = 0.25f * (float)(j % 7) - 0.75f;
    HIPCHK(hipMemcpyAsync(bias, hbias.data(), (size_t)Np * 4, hipMemcpyHostToDevice, e->stream));
    if (epilogue == 1) {
      hres.resize((size_t)Mp * Np);
      for (size_t i = 0; i < hres.size(); ++i) hres[i] = 0.125f * (float)((i * 2654435761u >> 7) % 33) - 2.0f;
      HIPCHK(hipMemcpyAsync(R, hres.data(), hres.size() * 4, hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  Bf16GemmArgs g;
  g.A = A; g.lda = K; g.B = B; g.ldb = K; g.M = M; g.N = N; g.K = K; g.kernel = kernel & (15 | 256 | 512); g.stagger = (kernel >> 4) & 15;
  g.tail = (kernel >> 10) & 15;   // (round 6) bits 10..13: tile variant of the tail launch (tail balancing, gemm_bf16.hip)
  EpiParams ep;
  ep.M = M; ep.N = N; ep.zero_pad = 1;
  int mode = EPI_STORE_F32;
  if (epilogue == 1) {            // bias + fp32 residual (to_out / fc2 shape of epilogue)
    mode = EPI_BIAS_RESID; ep.out = C; ep.ldo = Np; ep.resid = R; ep.ldr = Np; ep.bias = bias;
  } else if (epilogue == 2) {     // bias + GELU, two bf16 outputs (fc1)
    mode = EPI_BIAS_GELU; ep.out = C2; ep.ldo = Np; ep.out2 = C2 + Mp * Np; ep.ldo2 = Np; ep.bias = bias;
  } else if (epilogue == 3) {     // plain bf16 store (QKV / dgrads)
    mode = EPI_STORE; ep.out = C2; ep.ldo = Np;
  } else if (epilogue == 4) {     // fc2 input gradient: bf16 out = acc * stored gelu'(h), per-tile column sums (the fc1 bias gradient)
    mode = EPI_GELU_BWD; ep.out = C2; ep.ldo = Np; ep.aux = C2 + Mp * Np; ep.ldaux = Np; ep.colsum = C; ep.ldcs = Np;
    launch_fill_random_bf16(C2 + Mp * Np, Mp * Np, 3u, 1.0f, e->stream);
  } else {
    ep.out = C; ep.ldo = Np;
  }
  finalize_epi(ep);
  launch_gemm_bf16(g, ep, mode, e->stream);   // warm-up (+ attribute setup)
  unsigned long long* stamps = nullptr;
  if (vitx_env("VITX_GEMM_STAMPS")) {
    HIPCHK(hipMalloc((void**)&stamps, 512 * 16 * 4 * 8));
    HIPCHK(hipMemsetAsync(stamps, 0, 512 * 16 * 4 * 8, e->stream));
    g.stamps = stamps;
  }
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, e->stream));
  for (int i = 0; i < iters; ++i) launch_gemm_bf16(g, ep, mode, e->stream);
  HIPCHK(hipEventRecord(e1, e->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *avg_ms = ms / std::max(1, iters);
  if (stamps) {   // phase durations of the last launch, averaged over the workgroups, per tile index (cycles of the shader clock counter)
    std::vector<unsigned long long> hs(512 * 16 * 4);
    HIPCHK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    (void)hipFree(stamps);
    g.stamps = nullptr;
    if (atoi(vitx_env("VITX_GEMM_STAMPS")) == 2) {   // raw rows: workgroup, placement (xcc, HW_ID), then (tile start, K loop end, epilogue end, refill end) per tile
      for (int w = 0; w < 512; ++w) {
        const unsigned long long* p = &hs[(size_t)w * 16 * 4];
        if (!p[0]) continue;
        fprintf(stderr, "[stamps-raw] wg %d xcc %llu hwid 0x%08llx", w, p[15 * 4] >> 32, p[15 * 4] & 0xffffffffull);
        for (int t = 0; t < 15 && p[t * 4]; ++t) fprintf(stderr, " | %llu %llu %llu %llu", p[t * 4], p[t * 4 + 1], p[t * 4 + 2], p[t * 4 + 3]);
        fprintf(stderr, "\n");
      }
    }
    for (int t = 0; t < 15; ++t) {
      double kl = 0, ep_ = 0, dr = 0, gap = 0; int n = 0, ng = 0;
      for (int w = 0; w < 512; ++w) {
        const unsigned long long* p = &hs[((size_t)w * 16 + t) * 4];
        if (!p[0] || !p[3]) continue;
        kl += (double)(p[1] - p[0]); ep_ += (double)(p[2] - p[1]); dr += (double)(p[3] - p[2]); ++n;
        if (t + 1 < 15) { const unsigned long long* q = &hs[((size_t)w * 16 + t + 1) * 4]; if (q[0]) { gap += (double)(q[0] - p[3]); ++ng; } }
      }
      if (n) fprintf(stderr, "[stamps] tile %2d (%3d WGs): k-loop %8.0f  epilogue %8.0f  drain+refill %8.0f  gap %6.0f cycles\n", t, n, kl / n, ep_ / n, dr / n,
                     ng ? gap / ng : 0.0);
    }
  }
  *max_err = -1.f;
  if (check) {
    // reference: generic fp32-FMA kernel on the same operands; the fused epilogue is re-stated on the host
    float* C3;
    HIPCHK(hipMalloc((void**)&C3, (size_t)Mp * Np * 4));
    GenericGemmArgs gg;
    gg.A = A; gg.B = B; gg.M = M; gg.N = N; gg.K = K; gg.sam = K; gg.sak = 1; gg.sbk = 1; gg.sbn = K;
    EpiParams e2; e2.out = C3; e2.ldo = Np; e2.M = M; e2.N = N;
    finalize_epi(e2);
    launch_gemm_generic(gg, e2, EPI_STORE_F32, 1, 1, 0, e->stream);
    std::vector<float> ref((size_t)Mp * Np), got, got2;
    HIPCHK(hipMemcpyAsync(ref.data(), C3, ref.size() * 4, hipMemcpyDeviceToHost, e->stream));
    auto fetch_bf16 = [&](const bf16_t* src, std::vector<float>& dst) -> int {
      std::vector<uint16_t> raw((size_t)Mp * Np);
      HIPCHK(hipMemcpyAsync(raw.data(), src, raw.size() * 2, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      dst.resize(raw.size());
      for (size_t i = 0; i < raw.size(); ++i) { uint32_t u = (uint32_t)raw[i] << 16; float f; memcpy(&f, &u, 4); dst[i] = f; }
      return VITX_OK;
    };
    if (epilogue == 0 || epilogue == 1) {
      got.resize(ref.size());
      HIPCHK(hipMemcpyAsync(got.data(), C, got.size() * 4, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
    } else {
      if (int rc = fetch_bf16(C2, got)) return rc;
      if (epilogue == 2 || epilogue == 4) { if (int rc = fetch_bf16(C2 + Mp * Np, got2)) return rc; }
    }
    auto bf16r = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; memcpy(&f, &u, 4); return f; };
    float me = 0.f;
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) {
        const size_t o = (size_t)i * Np + j;
        float want = ref[o];
        if (epilogue == 1) want += hbias[j] + hres[o];
        if (epilogue == 2) want += hbias[j];
        float tol_scale = 1.f;
        if (epilogue >= 2) tol_scale = 1.f / (1.f + std::fabs(want));   // bf16 outputs: error relative to magnitude
        if (epilogue == 2) {
          // bf16 mode stores gelu'(h) and gelu(h), both evaluated on h as bf16 would round it (epilogue.h: kStoreGeluGrad).  A one-ulp
          // difference of that rounding between the two kernels' accumulation orders moves both by |dh| <= 2^-8 |h|: allow for it.
          const float h = bf16r(want);
          const float phi = 0.5f * (1.f + std::erf(h * 0.70710678f));
          const float gl = h * phi, gd = phi + h * 0.39894228f * std::exp(-0.5f * h * h);
          const float slack = 1.f / (1.f + 0.6f * std::fabs(h));
          me = std::max(me, std::fabs(got[o] - gd) * 0.5f * slack);
          me = std::max(me, std::fabs(got2[o] - gl) / (1.f + std::fabs(gl)) * slack);
        } else if (epilogue == 4) {
          want *= got2[o];
          me = std::max(me, std::fabs(got[o] - want) / (1.f + std::fabs(want)));
        } else {
          me = std::max(me, std::fabs(got[o] - want) * tol_scale);
        }
      }
    *max_err = me;
    (void)hipFree(C3);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(R); (void)hipFree(C2); (void)hipFree(bias);
  return VITX_OK;
}
