// Native data-parallel gradient exchange (SURVEY.md 8(b) "handle owns its ... RCCL communicator", 8(e), Appendix B): batch-sharded replicas, ONE
// exchange step per training step -- the mean of the gradient arena over the ranks -- done by the library itself:
//   * the arena is cut into fixed buckets (contiguous slices; the arena is laid out in reverse execution order of the backward, so buckets complete
//     front to back of the backward pass and no pack copy is needed);
//   * the engine reports arena ranges as their last producing kernel is enqueued (report_ready, engine.hip -- the same points the Python callback of
//     vit_tensorflow/parallel.py hangs on); the moment a bucket is fully covered, an event is recorded on the compute stream, the COMMUNICATION stream
//     waits for it and runs [fp32 -> bf16 wire copy] -> ncclAllReduce (RCCL, in place) -> [widen + x 1/world | x 1/world] for that bucket, so the
//     collective of block i runs under the backward of blocks i-1, i-2, ...;
//   * vitx_allreduce_grads() launches whatever was not reported, then makes the compute stream wait for the last bucket.
// Without vitx_comm_overlap the same entry point is the one-shot exchange behind the backward (every bucket at once).
// The reference has no distributed code: semantics are by equivalence -- N ranks x b/N images with mean-reduced gradients == one
// device on the concatenated batch.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are tens of MB so that RCCL spreads each one over all
// links / channels instead of paying per-message latency; the bf16 wire halves the bytes (173 instead of 346 MB at ViT-B/16).
// librccl is dlopen'ed (libvitx does not link against it); torch.distributed is NOT involved on this path.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>

#include "engine.h"

#define HIPCHK_ERR(x, err)                                                 \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      (err) = std::string(#x) + ": " + hipGetErrorString(e_);              \
      return VITX_ERR_HIP;                                                 \
    }                                                                      \
  } while (0)

namespace {

struct uid128_t { char b[128]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value to ncclCommInitRank
constexpr int NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9, NCCL_SUM = 0;   // ncclDataType_t / ncclRedOp_t values of RCCL's nccl.h
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);

// VITX_RCCL_LIB=<path>: the collective library to open instead of librccl.so (anything that exports ncclGetUniqueId / ncclCommInitRank / ncclAllReduce /
// ncclCommDestroy with RCCL's signatures).  tests/fake_rccl/ is such a library: a shared-memory stand-in that lets two ranks meet on a one-GPU box.
void* rccl_handle() {
  static void* lib = nullptr;
  if (lib) return lib;
  if (const char* p = vitx_env("VITX_RCCL_LIB")) {
    if (*p) {
      lib = dlopen(p, RTLD_NOW | RTLD_LOCAL);
      if (!lib) fprintf(stderr, "[vitx] VITX_RCCL_LIB=%s: %s\n", p, dlerror());
      return lib;   // (no silent fall-back to the system library: the caller asked for this one)
    }
  }
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  return lib;
}

__global__ __launch_bounds__(256) void comm_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = *(const float4*)(src + 4 * i);
    st4<bf16_t>(dst + 4 * i, v);
  }
}
__global__ __launch_bounds__(256) void comm_from_bf16_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n4, float alpha) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = ld4<bf16_t>(src + 4 * i);
    *(float4*)(dst + 4 * i) = make_float4(v.x * alpha, v.y * alpha, v.z * alpha, v.w * alpha);
  }
}
__global__ __launch_bounds__(256) void comm_scale_kernel(float* __restrict__ x, int64_t n4, float alpha) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 v = *(float4*)(x + 4 * i);
    *(float4*)(x + 4 * i) = make_float4(v.x * alpha, v.y * alpha, v.z * alpha, v.w * alpha);
  }
}
inline unsigned grid_for(int64_t n4) { return (unsigned)std::min<int64_t>(2048, std::max<int64_t>(1, (n4 + 255) / 256)); }

int64_t bucket_size(const vitx_engine* e, int i) { return std::min<int64_t>(e->cm.bucket, e->n_arena - (int64_t)i * e->cm.bucket); }

// bucket i is final on the compute stream: hand it to the communication stream
int launch_bucket(vitx_engine* e, int i, std::string& err) {
  CommState& c = e->cm;
  if (c.launched[(size_t)i]) return VITX_OK;
  c.launched[(size_t)i] = 1;
  auto ar = (all_reduce_fn)c.all_reduce;
  const int64_t lo = (int64_t)i * c.bucket, n = bucket_size(e, i);   // n_arena and the bucket size are multiples of 4 elements
  float* g = e->grads + lo;
  HIPCHK_ERR(hipEventRecord(c.ready_ev[(size_t)i], e->stream), err);
  HIPCHK_ERR(hipStreamWaitEvent(c.stream, c.ready_ev[(size_t)i], 0), err);
  // duration of this bucket's work two exchanges ago (its brackets have fired long since: the host runs at most one step ahead of the GPU)
  static const int timing_env = [] { const char* v = vitx_env("VITX_COMM_TIMING"); return v ? atoi(v) : 1; }();   // 0: no event brackets (A/B): every collective counts as unmeasured
  const int par = c.parity & 1;
  if (timing_env) {
    if (c.timed[par][(size_t)i] && hipEventQuery(c.t1_ev[par][(size_t)i]) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c.t0_ev[par][(size_t)i], c.t1_ev[par][(size_t)i]) == hipSuccess) c.coll_ms[(size_t)i] = ms;
    } else {
      (void)hipGetLastError();
    }
  }
  if (timing_env) HIPCHK_ERR(hipEventRecord(c.t0_ev[par][(size_t)i], c.stream), err);
  const float alpha = 1.0f / (float)e->world;
  if (c.wire_bf16) {
    bf16_t* w = c.wire + lo;
    hipLaunchKernelGGL(comm_to_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, c.stream, g, w, n / 4);
    if (ar(w, w, (size_t)n, NCCL_BFLOAT16, NCCL_SUM, e->comm, c.stream) != 0) { err = "ncclAllReduce (bf16 wire) failed"; return VITX_ERR_COMM; }
    hipLaunchKernelGGL(comm_from_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, c.stream, w, g, n / 4, alpha);
  } else {
    if (ar(g, g, (size_t)n, NCCL_FLOAT32, NCCL_SUM, e->comm, c.stream) != 0) { err = "ncclAllReduce failed"; return VITX_ERR_COMM; }
    if (e->world > 1) hipLaunchKernelGGL(comm_scale_kernel, dim3(grid_for(n / 4)), dim3(256), 0, c.stream, g, n / 4, alpha);
  }
  if (timing_env) {
    HIPCHK_ERR(hipEventRecord(c.t1_ev[par][(size_t)i], c.stream), err);
    c.timed[par][(size_t)i] = 1;
  }
  HIPCHK_ERR(hipEventRecord(c.done_ev[(size_t)i], c.stream), err);
  c.last_launched = i;
  ++c.n_launched;
  // the Dense launches enqueued from here on are the ones this collective runs beside: as many of them as its measured duration covers
  // (DENSE_US: a typical Dense launch of the backward pass; a collective not measured yet counts as long -- the conservative form, as before)
  constexpr float DENSE_US = 150.f;
  const float ms = c.coll_ms[(size_t)i];
  const int k = ms < 0.f ? 64 : (ms < 0.05f ? 0 : (int)(ms * 1000.f / DENSE_US) + 1);
  c.shared_credit = std::max(c.shared_credit, k);
  return VITX_OK;
}

int ensure_state(vitx_engine* e, std::string& err) {
  CommState& c = e->cm;
  if (!c.all_reduce) {
    c.all_reduce = dlsym(e->rccl_lib, "ncclAllReduce");
    if (!c.all_reduce) { err = "ncclAllReduce not found"; return VITX_ERR_COMM; }
  }
  if (!c.stream) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // (numerically lowest = highest priority): the few workgroups of a collective should not queue behind a GEMM's
    static const int prio_env = [] { const char* v = vitx_env("VITX_COMM_PRIORITY"); return v ? atoi(v) : 1; }();   // 1 highest (default), 0 default priority, -1 lowest
    HIPCHK_ERR(hipStreamCreateWithPriority(&c.stream, hipStreamNonBlocking, prio_env > 0 ? hi : (prio_env < 0 ? lo : 0)), err);
  }
  if (c.bucket <= 0) c.bucket = 8LL << 20;   // 32 MiB of fp32
  const int nb = (int)((e->n_arena + c.bucket - 1) / c.bucket);
  if ((int)c.launched.size() != nb) {
    for (auto ev : c.ready_ev) (void)hipEventDestroy(ev);
    for (auto ev : c.done_ev) (void)hipEventDestroy(ev);
    c.ready_ev.assign((size_t)nb, nullptr);
    c.done_ev.assign((size_t)nb, nullptr);
    for (int i = 0; i < nb; ++i) {
      HIPCHK_ERR(hipEventCreateWithFlags(&c.ready_ev[(size_t)i], hipEventDisableTiming), err);
      HIPCHK_ERR(hipEventCreateWithFlags(&c.done_ev[(size_t)i], hipEventDisableTiming), err);
    }
    for (int p = 0; p < 2; ++p) {
      for (auto ev : c.t0_ev[p]) (void)hipEventDestroy(ev);
      for (auto ev : c.t1_ev[p]) (void)hipEventDestroy(ev);
      c.t0_ev[p].assign((size_t)nb, nullptr);
      c.t1_ev[p].assign((size_t)nb, nullptr);
      c.timed[p].assign((size_t)nb, 0);
      for (int i = 0; i < nb; ++i) {
        HIPCHK_ERR(hipEventCreate(&c.t0_ev[p][(size_t)i]), err);
        HIPCHK_ERR(hipEventCreate(&c.t1_ev[p][(size_t)i]), err);
      }
    }
    c.coll_ms.assign((size_t)nb, -1.f);
    c.covered.assign((size_t)nb, 0);
    c.launched.assign((size_t)nb, 0);
    c.last_launched = -1;
    c.n_launched = 0;
    c.next_bucket = -1;
  }
  if (c.wire_bf16 && !c.wire) {
    HIPCHK_ERR(hipMalloc((void**)&c.wire, (size_t)e->n_arena * sizeof(bf16_t)), err);
  }
  return VITX_OK;
}

}  // namespace

int comm_unique_id(void* out128, std::string& err) {
  void* lib = rccl_handle();
  if (!lib) { err = "cannot dlopen librccl.so"; return VITX_ERR_COMM; }
  auto get = (int (*)(uid128_t*))dlsym(lib, "ncclGetUniqueId");
  if (!get) { err = "ncclGetUniqueId not found"; return VITX_ERR_COMM; }
  uid128_t id;
  if (get(&id) != 0) { err = "ncclGetUniqueId failed"; return VITX_ERR_COMM; }
  std::memcpy(out128, &id, 128);
  return VITX_OK;
}

int comm_init(vitx_engine* e, int rank, int world, const void* uid, std::string& err) {
  if (world < 1 || rank < 0 || rank >= world) { err = "vitx_comm_init: need 0 <= rank < world"; return VITX_ERR_INVALID; }
  if (e->comm) { err = "vitx_comm_init: this handle already has a communicator"; return VITX_ERR_STATE; }
  void* lib = rccl_handle();
  if (!lib) { err = "cannot dlopen librccl.so"; return VITX_ERR_COMM; }
  auto init = (int (*)(void**, int, uid128_t, int))dlsym(lib, "ncclCommInitRank");
  if (!init) { err = "ncclCommInitRank not found"; return VITX_ERR_COMM; }
  uid128_t id;
  std::memcpy(&id, uid, 128);
  HIPCHK_ERR(hipSetDevice(e->cfg.device_id), err);
  void* comm = nullptr;
  if (init(&comm, world, id, rank) != 0) { err = "ncclCommInitRank failed"; return VITX_ERR_COMM; }
  e->rccl_lib = lib;
  e->comm = comm;
  e->rank = rank;
  e->world = world;
  return VITX_OK;
}

// enable != 0: buckets are exchanged from inside the backward pass as they complete.  bucket_bytes: fp32 bytes per bucket (0 = 32 MiB); wire_bf16: the
// collective moves bf16 (rounded once per rank: 2^-9 relative per addend -- the usual price of a compressed gradient exchange; default fp32).
int comm_overlap(vitx_engine* e, int enable, int64_t bucket_bytes, int wire_bf16, std::string& err) {
  if (!e->comm) { err = "vitx_comm_init has not been called"; return VITX_ERR_STATE; }
  CommState& c = e->cm;
  if (c.n_launched) { err = "vitx_comm_overlap: an exchange is in progress (call vitx_allreduce_grads first)"; return VITX_ERR_STATE; }
  if (bucket_bytes < 0 || (bucket_bytes > 0 && bucket_bytes < 4096)) { err = "vitx_comm_overlap: bucket_bytes must be 0 (default) or >= 4096"; return VITX_ERR_INVALID; }
  const int64_t bucket = bucket_bytes > 0 ? (bucket_bytes / 4 + 3) / 4 * 4 : (8LL << 20);
  if (bucket != c.bucket) { c.bucket = bucket; c.launched.clear(); }   // (re-sized in ensure_state)
  c.wire_bf16 = wire_bf16 != 0;
  const int rc = ensure_state(e, err);
  if (rc != VITX_OK) return rc;
  std::fill(c.covered.begin(), c.covered.end(), 0);
  c.next_bucket = -1;
  c.overlap = enable != 0;
  return VITX_OK;
}

// engine.hip, report_ready: arena range [off, off + cnt) is final on the compute stream (every producing kernel has been enqueued there).
//  RCCL matches the collectives of a communicator by CALL ORDER, and every full bucket has the same element count: the order in which the
// buckets go out must not depend on what this rank's backward happened to run (CaiT layer dropout with per-rank seeds skips different blocks on
// different ranks).  Buckets therefore leave in ONE fixed order on every rank -- from the last bucket of the arena to the first, the order the
// backward completes them in -- and bucket i is launched only once it AND every bucket behind it are covered; a bucket covered early waits for
// its predecessors, a bucket never reported goes out from vitx_allreduce_grads, which continues the same descending order.
void comm_on_ready(vitx_engine* e, int64_t off, int64_t cnt) {
  CommState& c = e->cm;
  if (!c.overlap || c.launched.empty()) return;
  const int64_t lo = std::max<int64_t>(0, off), hi = std::min<int64_t>(e->n_arena, off + cnt);
  if (hi <= lo) return;
  std::string err;
  for (int64_t i = lo / c.bucket; i <= (hi - 1) / c.bucket; ++i) {
    const int64_t b0 = i * c.bucket, b1 = b0 + bucket_size(e, (int)i);
    if (c.launched[(size_t)i]) {   // this bucket has already gone out: a second backward pass ran before the exchange of the first was finished
      c.failed = "a backward pass ran before vitx_allreduce_grads finished the exchange of the previous one (one vitx_allreduce_grads per backward)";
      return;
    }
    c.covered[(size_t)i] += std::max<int64_t>(0, std::min(hi, b1) - std::max(lo, b0));
  }
  const int nb = (int)c.launched.size();
#ifdef VITX_COMM_LEGACY_ORDER   // (variant build only, tools/build_variant.sh: the round-5 rule -- a bucket leaves the moment it is covered -- to show that
                                //  tests/test_gpu_dp.py::test_ranks_with_different_layer_dropout_draws_exchange_the_same_buckets catches it)
  for (int i = 0; i < nb; ++i)
    if (!c.launched[(size_t)i] && c.covered[(size_t)i] >= bucket_size(e, i))
      if (launch_bucket(e, i, err) != VITX_OK) { c.failed = err; return; }
  return;
#endif
  if (c.next_bucket == -2) return;   // every bucket is out
  if (c.next_bucket < 0 || c.next_bucket >= nb) c.next_bucket = nb - 1;
  while (c.next_bucket >= 0 && !c.launched[(size_t)c.next_bucket] && c.covered[(size_t)c.next_bucket] >= bucket_size(e, c.next_bucket)) {
    if (launch_bucket(e, c.next_bucket, err) != VITX_OK) { c.failed = err; return; }
    --c.next_bucket;
    if (c.next_bucket < 0) { c.next_bucket = -2; break; }   // every bucket is out
  }
}

// the side stream's share of a reported arena range: every bucket launched from now on is ordered behind it (the communication stream is in-order)
void comm_wait_event(vitx_engine* e, hipEvent_t ev) {
  if (e->cm.stream) (void)hipStreamWaitEvent(e->cm.stream, ev, 0);
}

// 1 for the Dense launches that, in STREAM ORDER, run beside a collective of this handle: they use one-tile-per-workgroup grids (a persistent grid with
// static tile lists would wait for the workgroups RCCL's kernels keep off their CUs).  Round 5 asked the newest bucket's completion event -- at
// ENQUEUE time, tens of milliseconds before the launch executes, when it has practically never fired: every Dense launch behind the first bucket
// of a backward pass took the slower form (forced DP on one GPU: +0.8..1.0 ms).  Now each bucket, when it is launched, grants that form to as many
// following launches as its own measured duration (two exchanges ago) covers; a group of one, whose "collective" is a local copy, grants none.
int comm_busy(vitx_engine* e) {
  CommState& c = e->cm;
  if (!c.overlap || c.shared_credit <= 0) return 0;
  // VITX_COMM_SHARED=1 enables the rule; OFF by default since round 6: with the stub's emulated collectives (16-128 workgroups holding their CUs for
  // 0.3-1.5 ms per bucket, one rank, `profiles/r6/ab_emulated_collective_occupancy_rule_vs_persistent_r6ai.log`) persistent grids throughout are 0.1-0.5 ms
  // per step FASTER than the rule at every point but the widest / longest (64 x 1.5 ms: 38.20 vs 38.45) -- the late workgroups of a persistent launch
  // cost less than the slower one-tile forms of the 3-8 launches a bucket's duration covers.  Never measured against real multi-rank RCCL kernels.
  static const int honour = [] { const char* v = vitx_env("VITX_COMM_SHARED"); return v ? atoi(v) : 0; }();
  if (!honour) return 0;
  --c.shared_credit;
  ++c.busy_hits;
  return 1;
}

// vitx_allreduce_grads: finish (overlapped mode) or perform (one-shot mode) the exchange; on return the compute stream is ordered behind it
int comm_finish(vitx_engine* e, std::string& err) {
  if (!e->comm) { err = "vitx_comm_init has not been called"; return VITX_ERR_STATE; }
  int rc = ensure_state(e, err);
  if (rc != VITX_OK) return rc;
  CommState& c = e->cm;
  if (!c.failed.empty()) {
    err = c.failed;
    c.failed.clear();
    (void)hipStreamSynchronize(c.stream);
    std::fill(c.covered.begin(), c.covered.end(), 0);
    std::fill(c.launched.begin(), c.launched.end(), 0);
    c.last_launched = -1;
    c.n_launched = 0;
    c.next_bucket = -1;
    return VITX_ERR_COMM;
  }
  const int nb = (int)c.launched.size();
  c.last_overlapped = c.n_launched;   // buckets that went out from inside the backward pass
  for (int i = nb - 1; i >= 0; --i)   // the same fixed order as comm_on_ready: last bucket first
    if ((rc = launch_bucket(e, i, err)) != VITX_OK) return rc;
  // the communication stream runs its buckets in order: the newest completion event covers them all
  if (c.last_launched >= 0) HIPCHK_ERR(hipStreamWaitEvent(e->stream, c.done_ev[(size_t)c.last_launched], 0), err);
  c.parity ^= 1;            // the next exchange records into the other set of brackets and reads this one's the time after
  c.shared_credit = 0;      // the compute stream is now ordered behind every collective of this exchange
  std::fill(c.covered.begin(), c.covered.end(), 0);
  std::fill(c.launched.begin(), c.launched.end(), 0);
  c.last_launched = -1;
  c.n_launched = 0;
  c.next_bucket = -1;
  return VITX_OK;
}

// {buckets, buckets the last exchange sent from inside the backward pass, elements per bucket, Dense launches that found a collective in flight}
void comm_stats(vitx_engine* e, int64_t* out4) {
  const CommState& c = e->cm;
  out4[0] = (int64_t)c.launched.size();
  out4[1] = c.last_overlapped;
  out4[2] = c.bucket;
  out4[3] = c.busy_hits;
}

void comm_destroy(vitx_engine* e) {
  CommState& c = e->cm;
  if (c.stream) (void)hipStreamSynchronize(c.stream);
  for (auto ev : c.ready_ev) (void)hipEventDestroy(ev);
  for (auto ev : c.done_ev) (void)hipEventDestroy(ev);
  c.ready_ev.clear();
  c.done_ev.clear();
  for (int p = 0; p < 2; ++p) {
    for (auto ev : c.t0_ev[p]) (void)hipEventDestroy(ev);
    for (auto ev : c.t1_ev[p]) (void)hipEventDestroy(ev);
    c.t0_ev[p].clear(); c.t1_ev[p].clear(); c.timed[p].clear();
  }
  c.coll_ms.clear();
  c.shared_credit = 0;
  if (c.wire) (void)hipFree(c.wire);
  c.wire = nullptr;
  if (c.stream) (void)hipStreamDestroy(c.stream);
  c.stream = nullptr;
  if (e->comm && e->rccl_lib) {
    auto destroy = (int (*)(void*))dlsym(e->rccl_lib, "ncclCommDestroy");
    if (destroy) (void)destroy(e->comm);
  }
  e->comm = nullptr;
  e->rank = 0;
  e->world = 1;
  // single-rank again: no bucket state, no launches from inside the backward (vitx_comm_destroy between steps)
  c.overlap = 0;
  c.covered.clear();
  c.launched.clear();
  c.last_launched = -1;
  c.n_launched = 0;
  c.next_bucket = -1;
  c.failed.clear();
}
