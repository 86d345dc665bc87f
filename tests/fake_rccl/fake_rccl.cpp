// TEST INFRASTRUCTURE (not product code): a stand-in for the four librccl entry points csrc/comm.hip binds, so that TWO RANKS CAN MEET ON A ONE-GPU BOX.
//
// The pool this repository is developed on has one MI355X per box; RCCL refuses two ranks on the same device.  csrc/comm.hip opens its collective
// library by name (`VITX_RCCL_LIB=<path>` overrides `librccl.so`), so the tests point it at this library instead and run the data-parallel exchange
// of the engine -- bucket geometry, bucket ORDER across ranks, the gradient-ready ordering against the weight-gradient stream, the 1/world average,
// the one-tile-per-workgroup GEMM switch while a collective is in flight -- as two processes on GPU 0 (tests/test_gpu_dp.py).
//
// What it implements, with RCCL's signatures (nccl.h of ROCm 7.2):
//   ncclGetUniqueId      a name for a POSIX shared-memory segment
//   ncclCommInitRank     maps the segment (header + one staging slot per rank), waits for every rank
//   ncclAllReduce        STREAM-ORDERED like the real one: device -> this rank's slot (hipMemcpyAsync) -> a host function on the stream that waits for
//                        every rank's slot of the same collective index, checks that all ranks posted the SAME element count and dtype (RCCL matches
//                        collectives by call order: a rank that sends its buckets in another order is caught here instead of summing mismatched data),
//                        sums the slots in rank order (fp32 accumulation; bf16 wire rounded to nearest even) -> host -> device (hipMemcpyAsync)
//   ncclCommDestroy      unmaps, the last rank unlinks the segment
// Optional CU occupancy (VITX_FAKE_RCCL_SPIN_WGS=n, VITX_FAKE_RCCL_SPIN_US=t): before its reduction every collective runs a kernel of n workgroups
// (256 threads, 96 VGPRs each: no persistent GEMM workgroup fits beside one) that spin for t microseconds -- what RCCL's channel workgroups do to a
// compute stream for the length of a collective.  With one rank this emulates, on a one-GPU box, the situation the engine's one-tile-per-workgroup
// GEMM forms exist for (profiles/r6/ab_emulated_collective_occupancy_*.log).
// Every wait has a deadline (VITX_FAKE_RCCL_TIMEOUT_S, default 120 s): a rank that never arrives turns into an error result (NaN-filled output and a
// non-zero return from the next call), not a hang.  Sum only (ncclSum), float32 / bfloat16 only -- what comm.hip uses.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

// holds `gridDim.x` CUs' worth of registers for `ticks` periods of the 100-MHz wall clock
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks) {
  asm volatile("v_mov_b32 v95, 0" ::: "v95");
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

__global__ void empty_kernel() {}

constexpr int MAX_RANKS = 8;
constexpr int NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9, NCCL_SUM = 0;
constexpr int ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4;

struct Header {
  std::atomic<int> arrived;               // ranks that have mapped the segment
  std::atomic<int> left;                  // ranks that have destroyed their communicator
  std::atomic<int> error;                 // sticky: mismatched collectives / a rank that did not arrive in time
  std::atomic<uint64_t> posted[MAX_RANKS];     // collectives whose input is in the rank's slot
  std::atomic<uint64_t> consumed[MAX_RANKS];   // collectives whose slots the rank has finished reading
  std::atomic<uint64_t> count[MAX_RANKS];      // element count and dtype of the rank's newest posted collective
  std::atomic<int> dtype[MAX_RANKS];
};
constexpr size_t HEADER_BYTES = 4096;
static_assert(sizeof(Header) <= HEADER_BYTES, "header page");

struct Comm {
  int rank = 0, world = 1;
  char name[96] = {0};
  char* map = nullptr;
  size_t map_bytes = 0, slot_bytes = 0;
  Header* hdr = nullptr;
  char* result = nullptr;        // pinned host buffer of one slot: the reduced values on their way back to the device
  uint64_t calls = 0;            // collectives enqueued by this rank (host side)
  bool registered = false;
  double timeout_s = 120.0;
  int spin_wgs = 0;              // emulated channel workgroups per collective (0: none)
  double spin_us = 0.0;
};

struct Call { Comm* c; uint64_t k; size_t count; int dtype; };

inline float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
inline uint16_t f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <typename F>
bool wait_until(const Comm* c, F&& cond) {
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (!cond()) {
    if (c->hdr->error.load(std::memory_order_acquire)) return false;
    if (++spins > 2000) {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) return false;
    }
  }
  return true;
}

void poison(Comm* c, size_t count, int dtype) {   // an error result nobody can mistake for a gradient
  if (dtype == NCCL_FLOAT32) { float* r = (float*)c->result; for (size_t i = 0; i < count; ++i) r[i] = NAN; }
  else { uint16_t* r = (uint16_t*)c->result; for (size_t i = 0; i < count; ++i) r[i] = 0x7fc0; }
}

// runs on the stream, after this rank's input has landed in its slot
void reduce_on_host(void* user) {
  Call* call = (Call*)user;
  Comm* c = call->c;
  Header* h = c->hdr;
  const uint64_t k1 = call->k + 1;
  h->count[c->rank].store(call->count, std::memory_order_relaxed);
  h->dtype[c->rank].store(call->dtype, std::memory_order_relaxed);
  h->posted[c->rank].store(k1, std::memory_order_release);
  bool ok = wait_until(c, [&] { for (int r = 0; r < c->world; ++r) if (h->posted[r].load(std::memory_order_acquire) < k1) return false; return true; });
  if (ok) {
    // RCCL pairs the k-th collective of every rank: they must describe the same buffer
    for (int r = 0; r < c->world && ok; ++r) {
      // (a rank may already have posted collective k + 1 only after everyone consumed k: the count read here still belongs to k)
      if (h->count[r].load(std::memory_order_relaxed) != call->count || h->dtype[r].load(std::memory_order_relaxed) != call->dtype) {
        fprintf(stderr, "[fake_rccl] rank %d: collective %llu is %zu elements of type %d here but %llu of type %d on rank %d -- the ranks launch their "
                        "collectives in different orders\n", c->rank, (unsigned long long)call->k, call->count, call->dtype,
                (unsigned long long)h->count[r].load(), h->dtype[r].load(), r);
        ok = false;
      }
    }
  } else if (!h->error.load()) {
    fprintf(stderr, "[fake_rccl] rank %d: collective %llu timed out waiting for the other ranks\n", c->rank, (unsigned long long)call->k);
  }
  if (!ok) {
    h->error.store(1, std::memory_order_release);
    poison(c, call->count, call->dtype);
    delete call;
    return;
  }
  const char* slots = c->map + HEADER_BYTES;
  if (call->dtype == NCCL_FLOAT32) {
    float* out = (float*)c->result;
    for (size_t i = 0; i < call->count; ++i) {
      float a = ((const float*)slots)[i];
      for (int r = 1; r < c->world; ++r) a += ((const float*)(slots + (size_t)r * c->slot_bytes))[i];   // rank order: the same bits on every rank
      out[i] = a;
    }
  } else {
    uint16_t* out = (uint16_t*)c->result;
    for (size_t i = 0; i < call->count; ++i) {
      float a = bf16_to_f32(((const uint16_t*)slots)[i]);
      for (int r = 1; r < c->world; ++r) a += bf16_to_f32(((const uint16_t*)(slots + (size_t)r * c->slot_bytes))[i]);
      out[i] = f32_to_bf16(a);
    }
  }
  h->consumed[c->rank].store(k1, std::memory_order_release);
  // nobody may overwrite a slot (the next collective's device -> host copy) before every rank has read it
  if (!wait_until(c, [&] { for (int r = 0; r < c->world; ++r) if (h->consumed[r].load(std::memory_order_acquire) < k1) return false; return true; })) {
    h->error.store(1, std::memory_order_release);
    poison(c, call->count, call->dtype);
  }
  delete call;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void* out128) {
  char name[128];
  memset(name, 0, sizeof name);
  snprintf(name, sizeof name, "/vitx_fake_rccl_%d_%lld", (int)getpid(), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  memcpy(out128, name, 128);
  return ncclSuccess;
}

struct uid128 { char b[128]; };

int ncclCommInitRank(void** comm_out, int nranks, uid128 id, int rank) {
  if (!comm_out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm;
  c->rank = rank; c->world = nranks;
  memcpy(c->name, id.b, sizeof c->name - 1);
  if (const char* t = getenv("VITX_FAKE_RCCL_TIMEOUT_S")) c->timeout_s = atof(t);
  if (const char* t = getenv("VITX_FAKE_RCCL_SPIN_WGS")) c->spin_wgs = atoi(t);
  if (const char* t = getenv("VITX_FAKE_RCCL_SPIN_US")) c->spin_us = atof(t);
  const char* mb = getenv("VITX_FAKE_RCCL_SLOT_MB");
  c->slot_bytes = (size_t)(mb ? atoi(mb) : 64) << 20;
  c->map_bytes = HEADER_BYTES + (size_t)nranks * c->slot_bytes;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { perror("[fake_rccl] shm_open"); delete c; return ncclSystemError; }
  if (ftruncate(fd, (off_t)c->map_bytes) != 0) { perror("[fake_rccl] ftruncate"); close(fd); delete c; return ncclSystemError; }   // (fresh pages are zero: the header needs no init)
  c->map = (char*)mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->map == MAP_FAILED) { perror("[fake_rccl] mmap"); delete c; return ncclSystemError; }
  c->hdr = (Header*)c->map;
  if (hipHostMalloc((void**)&c->result, c->slot_bytes, hipHostMallocDefault) != hipSuccess) { munmap(c->map, c->map_bytes); delete c; return ncclSystemError; }
  // pinned slots make the device -> host copies truly asynchronous; pageable ones are still stream-ordered (the copy then blocks the calling thread)
  c->registered = hipHostRegister(c->map + HEADER_BYTES + (size_t)rank * c->slot_bytes, c->slot_bytes, hipHostRegisterDefault) == hipSuccess;
  if (!c->registered) (void)hipGetLastError();
  if (const char* t = getenv("VITX_FAKE_RCCL_DUMMY_STREAMS")) {   // experiment: what real RCCL's own streams do to the stream -> hardware-queue mapping
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int i = 0; i < atoi(t); ++i) {
      hipStream_t st;
      const int pr = getenv("VITX_FAKE_RCCL_DUMMY_PRIO") ? atoi(getenv("VITX_FAKE_RCCL_DUMMY_PRIO")) : 0;
      if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr > 0 ? hi : (pr < 0 ? lo : 0)) == hipSuccess) {
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st);
        (void)hipStreamSynchronize(st);
      }
    }
  }
  c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel);
  if (!wait_until(c, [&] { return c->hdr->arrived.load(std::memory_order_acquire) >= nranks; })) {
    fprintf(stderr, "[fake_rccl] rank %d: only %d of %d ranks arrived\n", rank, c->hdr->arrived.load(), nranks);
    return ncclSystemError;
  }
  *comm_out = c;
  return ncclSuccess;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || op != NCCL_SUM || (dtype != NCCL_FLOAT32 && dtype != NCCL_BFLOAT16)) return ncclInvalidArgument;
  const size_t bytes = count * (dtype == NCCL_FLOAT32 ? 4 : 2);
  if (bytes > c->slot_bytes) { fprintf(stderr, "[fake_rccl] %zu bytes per collective exceed the slot (VITX_FAKE_RCCL_SLOT_MB)\n", bytes); return ncclInvalidArgument; }
  if (c->hdr->error.load(std::memory_order_acquire)) return ncclSystemError;
  char* slot = c->map + HEADER_BYTES + (size_t)c->rank * c->slot_bytes;
  if (c->spin_wgs > 0 && c->spin_us > 0.0) {
    static const int mode = [] { const char* v = getenv("VITX_FAKE_RCCL_SPIN_MODE"); return v ? atoi(v) : 0; }();   // experiments: 1 = no kernel at all, 2 = an empty kernel
    if (mode == 0) hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)c->spin_wgs), dim3(256), 0, stream, (unsigned long long)(c->spin_us * 100.0));
    else if (mode == 2) hipLaunchKernelGGL(empty_kernel, dim3((unsigned)c->spin_wgs), dim3(256), 0, stream);
    if (c->world == 1) {   // occupancy emulation with one rank: the sum over one rank is the identity -- no trip through host memory, only the held CUs
      if (send != recv && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclSystemError;
      return ncclSuccess;
    }
  }
  if (hipMemcpyAsync(slot, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclSystemError;
  Call* call = new Call{c, c->calls++, count, dtype};
  if (hipLaunchHostFunc(stream, reduce_on_host, call) != hipSuccess) { delete call; return ncclSystemError; }
  if (hipMemcpyAsync(recv, c->result, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclSystemError;
  return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclSuccess;
  if (c->registered) (void)hipHostUnregister(c->map + HEADER_BYTES + (size_t)c->rank * c->slot_bytes);
  const int left = c->hdr->left.fetch_add(1, std::memory_order_acq_rel) + 1;
  if (left >= c->world) shm_unlink(c->name);
  munmap(c->map, c->map_bytes);
  (void)hipHostFree(c->result);
  delete c;
  return ncclSuccess;
}

}  // extern "C"
