// Per-workgroup timeline of the fused bf16 attention kernels (vit.py:73-82 and its VJP) at the benchmarked shape: the product source
// attn_bf16.hip compiled with -DVITX_ATTN_PROBE records 100-MHz timestamps at the phase boundaries of wave 0 of every workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -DVITX_ATTN_PROBE -I vit-tensorflow_amd/csrc -I include tools/probe_attn.hip -o tools/probe_attn
//   tools/probe_attn [batch=256] [tokens=197] [heads=12] [planar=0] [threads=512]     planar 1: q | k | v / o / dO as [plane][b n][64] planes (AttnLayout 1), the
//                                                                        A/B of the access pattern (contiguous n x 128-B blocks per task)
#include "../vit-tensorflow_amd/csrc/attn_bf16.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void report(const char* what, const std::vector<unsigned long long>& st, int nwg, const char* const* names, int nphase) {
  // ticks are 10 ns
  std::vector<double> sum(nphase, 0.0);
  unsigned long long first = ~0ull, last = 0;
  std::vector<double> life(nwg);
  for (int w = 0; w < nwg; ++w) {
    const unsigned long long* s = &st[(size_t)w * 8];
    for (int p = 0; p < nphase; ++p) sum[p] += (double)(s[p + 1] - s[p]);
    first = std::min(first, s[0]);
    last = std::max(last, s[nphase]);
    life[w] = (double)(s[nphase] - s[0]) * 0.01;
  }
  std::sort(life.begin(), life.end());
  printf("%s: %d workgroups, first start -> last end %.1f us; workgroup lifetime (wave 0) mean %.2f us  p10 %.2f  p50 %.2f  p90 %.2f\n", what, nwg,
         (double)(last - first) * 0.01, [&] { double a = 0; for (double x : life) a += x; return a / nwg; }(), life[nwg / 10], life[nwg / 2], life[nwg * 9 / 10]);
  for (int p = 0; p < nphase; ++p) printf("    %-46s %6.2f us\n", names[p], sum[p] / nwg * 0.01);
  {   // ramp: when does the k-th workgroup start?  (two per CU = 512 resident at once if nothing but the slots limits it)
    std::vector<unsigned long long> starts(nwg);
    for (int w = 0; w < nwg; ++w) starts[w] = st[(size_t)w * 8];
    std::sort(starts.begin(), starts.end());
    printf("    start of workgroup #k after the first:");
    for (int k : {64, 128, 255, 256, 257, 320, 384, 511, 512, 768, 1024}) if (k < nwg) printf("  #%d %.2f us", k, (double)(starts[k] - starts[0]) * 0.01);
    printf("\n");
  }
  // residency per CU: HW_ID bits cu 11:8, sh 12, se 15:13 (+ XCC_ID 3:0 in the upper word)
  std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
  for (int w = 0; w < nwg; ++w) {
    const unsigned long long id = st[(size_t)w * 8 + 6];
    const unsigned key = (unsigned)((id >> 8) & 0xff) | ((unsigned)((id >> 32) & 0xf) << 8);
    cu[key].push_back({st[(size_t)w * 8], st[(size_t)w * 8 + nphase]});
  }
  double busy1 = 0, busy2 = 0, span = 0, gap = 0; long ngap = 0;
  for (auto& kv : cu) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end());
    std::vector<std::pair<unsigned long long, int>> ev;
    for (auto& iv : v) { ev.push_back({iv.first, 1}); ev.push_back({iv.second, -1}); }
    std::sort(ev.begin(), ev.end());
    int depth = 0; unsigned long long prev = ev.front().first;
    for (auto& e : ev) {
      const double dt = (double)(e.first - prev) * 0.01;
      if (depth == 1) busy1 += dt; else if (depth >= 2) busy2 += dt; else if (dt > 0) { gap += dt; ++ngap; }
      depth += e.second; prev = e.first;
    }
    span += (double)(ev.back().first - ev.front().first) * 0.01;
  }
  printf("    %zu compute units seen, %.1f workgroups each; per CU: %.1f us with one workgroup resident, %.1f us with two or more, %.1f us with none (%.1f gaps)\n",
         cu.size(), (double)nwg / cu.size(), busy1 / cu.size(), busy2 / cu.size(), gap / cu.size(), (double)ngap / cu.size());
}

int main(int argc, char** argv) {
  const int b = argc > 1 ? atoi(argv[1]) : 256, n = argc > 2 ? atoi(argv[2]) : 197, h = argc > 3 ? atoi(argv[3]) : 12;
  vitx_attn_probe_planar = argc > 4 ? atoi(argv[4]) : 0;
  vitx_attn_probe_threads = argc > 5 ? atoi(argv[5]) : 512;
  printf("%d threads per workgroup\n", vitx_attn_probe_threads);
  printf("batch %d, %d tokens, %d heads, layout %s\n", b, n, h, vitx_attn_probe_planar ? "planar [3h][b n][64]" : "interleaved [b n][3][h][64] (the engine's)");
  const size_t tok = (size_t)b * n, inner = (size_t)h * 64;
  std::vector<bf16_t> hq(tok * 3 * inner), hd(tok * inner);
  srand(1);
  for (auto& v : hq) v = (bf16_t)(((rand() & 1023) - 512) / 512.0f);
  for (auto& v : hd) v = (bf16_t)(((rand() & 1023) - 512) / 4096.0f);
  bf16_t *qkv, *o, *d_o, *dqkv, *zero;
  float *lse, *dsum;
  unsigned long long* stamps;
  CK(hipMalloc(&qkv, hq.size() * 2)); CK(hipMalloc(&dqkv, hq.size() * 2)); CK(hipMalloc(&o, hd.size() * 2)); CK(hipMalloc(&d_o, hd.size() * 2));
  CK(hipMalloc(&lse, (size_t)b * h * n * 4)); CK(hipMalloc(&dsum, (size_t)b * h * n * 4)); CK(hipMalloc(&zero, 4096)); CK(hipMemset(zero, 0, 4096));
  const int nwg = b * h;
  CK(hipMalloc(&stamps, (size_t)nwg * 64)); CK(hipMemset(stamps, 0, (size_t)nwg * 64));
  CK(hipMemcpy(qkv, hq.data(), hq.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_o, hd.data(), hd.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(vitx_attn_probe_buf), &stamps, sizeof(stamps)));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> st((size_t)nwg * 8);
  const float scale = 0.125f;
  for (int pass = 0; pass < 2; ++pass) {
    float ms = 0.f;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0, s));
      if (pass == 0) launch_attn_bf16_fwd(qkv, o, lse, b, n, h, scale, zero, 0, s);
      else launch_attn_bf16_bwd(qkv, o, d_o, lse, dsum, dqkv, b, n, h, scale, zero, s);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    printf("launch %.1f us (with the stamps)\n", ms * 1000.f);
    static const char* const fn[] = {"issue staging + q loads, wait for own DMA", "workgroup barrier", "pass 1 (row maxima)", "pass 2 (exp, sums, P V)", "normalise + store"};
    static const char* const bn[] = {"wait for K / V images", "phase 1 (dQ, D)", "barrier (slowest wave of phase 1)", "stage Q / dO, wait", "phase 2 (dK, dV)"};
    report(pass == 0 ? "forward" : "backward (fused)", st, nwg, pass == 0 ? fn : bn, 5);
  }
  return 0;
}
