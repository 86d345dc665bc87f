// The registry behind env.h: every VITX_* variable the library reads, its class and what it does.  INTEGRATION.md section 3 is generated from this
// table (tools/env_table.py) and tests/test_abi.py checks that the two agree and that a release build ignores the DIAG class.
#include "env.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>

namespace {

const VitxEnvSwitch kSwitches[] = {
    // ---- code paths with their own tests (results within the same oracle gates)
    {"VITX_GENERIC_GEMM", VITX_ENV_PATH, "1: every Dense on the k-ordered fp32-FMA kernel instead of the MFMA kernels"},
    {"VITX_GENERIC_ATTN", VITX_ENV_PATH, "1: materialised attention (batched products + row softmax) instead of the fused kernel"},
    {"VITX_DEEPVIT_FUSED", VITX_ENV_PATH, "0: DeepViT Re-attention forward as batched GEMMs + head-axis kernels instead of the one-kernel form"},
    {"VITX_DEEPVIT_FUSED_BWD", VITX_ENV_PATH, "0: DeepViT Re-attention backward as batched GEMMs + point / row kernels"},
    {"VITX_CAIT_FUSED", VITX_ENV_PATH, "0: CaiT talking-heads forward as batched GEMMs + chain kernel instead of the one-kernel form"},
    {"VITX_CAIT_QKV_CAT", VITX_ENV_PATH, "0: CaiT patch stage: to_q and to_kv as two Dense launches each way (rounds d(y1) twice)"},
    {"VITX_LN_SCALE_FUSED", VITX_ENV_PATH, "0: CaiT LayerScale VJP as a pass of its own instead of on the LayerNorm VJP (bias gradient from the bf16 rounding)"},
    {"VITX_CHAIN_MFMA", VITX_ENV_PATH, "0: talking-heads chain kernels with in-lane FMAs instead of the fp32 matrix pipe"},
    {"VITX_UNFUSED_HEADOPS", VITX_ENV_PATH, "1: head-axis operations (mix / softmax / LayerNorm over heads) as one kernel each"},
    {"VITX_WGRAD_TRANSPOSE", VITX_ENV_PATH, "1: weight gradients through explicit bf16 transposes + the NT kernel"},
    {"VITX_GELU_TABLE", VITX_ENV_PATH, "0: fc1 epilogue evaluates the degree-7 polynomial GELU instead of the LDS table (another rounding of gelu, both inside a bf16 ulp)"},
    {"VITX_X3_ATTN", VITX_ENV_PATH, "BF16X3 handles: 1 fused split-operand attention (default), 2 materialised scores + split-operand products, 0 exact fp32 products"},
    {"VITX_F32_MFMA", VITX_ENV_TUNING, "0: parity-mode GEMMs on the scalar-FMA kernel instead of the fp32 matrix pipe (same bits)"},
    // ---- tuning: same results
    {"VITX_GEMM_KERNEL", VITX_ENV_TUNING, "force one NT tile variant (1, 2, 3, 5, 6, 7, 9, 10, 11, 13; +256 direct / +512 LDS-staged epilogue); every variant adds in the same K order"},
    {"VITX_GEMM_TAIL_KERNEL", VITX_ENV_TUNING, "with VITX_GEMM_KERNEL=13 / 11: tile variant (1, 3, 10) of the tail launch (tail balancing)"},
    {"VITX_GEMM_TAIL", VITX_ENV_TUNING, "1: the per-shape measurement also times the tail-balanced forms (off by default: loses inside the step)"},
    {"VITX_GEMM_AUTOTUNE", VITX_ENV_TUNING, "0: static variant rule instead of the first-launch measurement"},
    {"VITX_GEMM_AUTOTUNE_LOG", VITX_ENV_TUNING, "1: print the measured variant choices"},
    {"VITX_GEMM_WALK", VITX_ENV_TUNING, "0: no XCD-owned row bands in the persistent NT kernel"},
    {"VITX_GEMM_GRID", VITX_ENV_TUNING, "n: persistent workgroups per launch (default 256)"},
    {"VITX_GEMM_PHASE", VITX_ENV_TUNING, "n: start-phase offsets between the persistent workgroups of an XCD (sleeps only)"},
    {"VITX_EPI_WIDE", VITX_ENV_TUNING, "0: 4-B instead of 16-B-per-lane epilogue accesses"},
    {"VITX_NT", VITX_ENV_TUNING, "bits: non-temporal stores (1 = gelu' of the fc1 epilogue, 16 = d(y) of the fc1 / qkv input gradients; default 17)"},
    {"VITX_REVERSE", VITX_ENV_TUNING, "bits: row tiles walked back to front for operands larger than the memory-side cache (1 forward, 2 input gradients)"},
    {"VITX_REVERSE_MIN_MB", VITX_ENV_TUNING, "operand size from which VITX_REVERSE applies"},
    {"VITX_MLP_BWD_ORDER", VITX_ENV_TUNING, "0: fc2 weight gradient between the producer and the consumers of d(hpre)"},
    {"VITX_BGEMM_PAIRS", VITX_ENV_TUNING, "0: the batched products of the materialised attention backward as four launches instead of two (same bits)"},
    {"VITX_ATTN_BWD_SPLIT", VITX_ENV_TUNING, "1: fused attention backward as two launches (same bits)"},
    {"VITX_GLP_SKIP", VITX_ENV_TUNING, "0: LayerNorm VJPs always write the bf16 copy of the residual gradient"},
    {"VITX_SCORE_BF16", VITX_ENV_TUNING, "0: DeepViT / CaiT score tensors that only batched products read stay fp32 planes (same bits)"},
    {"VITX_LN_CORESIDENT", VITX_ENV_TUNING, "1: LayerNorm VJP of 768-wide bf16 rows as 4-wave, 96-register blocks that fit beside a resident weight-gradient GEMM (experiment; partials summed in another order)"},
    {"VITX_LN_CO_BLOCKS", VITX_ENV_TUNING, "blocks of the co-resident LayerNorm VJP form (default 1024; 256 = one 4-wave block per CU, what fits beside a resident GEMM workgroup)"},
    {"VITX_SIDE_STREAM", VITX_ENV_TUNING, "0 one stream; 1 weight gradients on a lowest-priority side stream (default); 2 the same at the highest priority"},
    {"VITX_LN_REDUCE_SIDE_ROWS", VITX_ENV_TUNING, "LayerNorm VJPs of fewer rows keep their small reductions on the main stream (default 8192)"},
    {"VITX_SC_KEEP_MB", VITX_ENV_TUNING, "budget of the score tensors kept per block for the backward; blocks beyond it recompute (same bits)"},
    {"VITX_RECOMPUTE_SCORES", VITX_ENV_TUNING, "1: materialised attention recomputes its score tensors in the backward (same bits)"},
    {"VITX_WGRAD_WGS", VITX_ENV_TUNING, "workgroups per weight-gradient launch (default 128 up to 24576 token rows, else 256); changes the split-K slice count: fixed-order sums in another order"},
    {"VITX_SG_BLOCKS", VITX_ENV_TUNING, "blocks per LayerScale-gradient launch"},
    {"VITX_DV_CPI", VITX_ENV_TUNING, "chunks per image of the DeepViT one-kernel forward (tests: the multi-tile loop at small batches)"},
    {"VITX_COMM_PRIORITY", VITX_ENV_TUNING, "priority of the communication stream: 1 highest (default), 0 default, -1 lowest"},
    {"VITX_COMM_TIMING", VITX_ENV_TUNING, "0: no event brackets around the collectives (every Dense launch behind a bucket then takes the one-tile form)"},
    {"VITX_COMM_WAIT_ON_CHAIN", VITX_ENV_TUNING, "1: the compute stream (not the communication stream) waits for the weight-gradient stream's share of a reported range"},
    {"VITX_COMM_STREAM_EARLY", VITX_ENV_TUNING, "0: the communication stream is created at vitx_comm_overlap instead of with the handle's other streams (A/B: it may then share a command-processor pipe with the compute stream)"},
    {"VITX_COMM_SHARED", VITX_ENV_TUNING, "1: the Dense launches behind a bucket take the one-tile-per-workgroup form for as long as its collective was measured to last (default 0: persistent grids throughout, faster with emulated collectives)"},
    {"VITX_RCCL_LIB", VITX_ENV_TUNING, "path of the collective library to dlopen instead of librccl.so (tests/fake_rccl: two ranks on one GPU)"},
    // ---- diagnostics: may corrupt results; diagnostic build only
    {"VITX_GEMM_STAGGER", VITX_ENV_DIAG, "bits reach the NT kernels' experiment field: 1 no DMA wait, 2 no DMA issue (WRONG GEMM results)"},
    {"VITX_GEMM_XP", VITX_ENV_DIAG, "1: vitx_bench_gemm accepts the NT kernels' experiment bits (results invalid)"},
    {"VITX_TN_XP", VITX_ENV_DIAG, "bits: weight-gradient kernel without DMA wait / DMA issue / fragment reads (WRONG weight gradients)"},
    {"VITX_DV_XP", VITX_ENV_DIAG, "bits: DeepViT one-kernel forward without kept tensors / stage 2 / stage 3 (WRONG results)"},
    {"VITX_DVB_XP", VITX_ENV_DIAG, "bits: DeepViT one-kernel backward without one of its stages (WRONG results)"},
    {"VITX_BENCH_ZERO", VITX_ENV_DIAG, "vitx_bench_gemm on zero operands (DVFS / power-limit experiment)"},
    {"VITX_GEMM_STAMPS", VITX_ENV_DIAG, "vitx_bench_gemm prints the cycle stamps of the tile phases (needs the stamps build of gemm_bf16_pipe.hip)"},
};
constexpr int kNumSwitches = (int)(sizeof(kSwitches) / sizeof(kSwitches[0]));

#ifdef VITX_DIAG
constexpr int kDiagBuild = 1;
#else
constexpr int kDiagBuild = 0;
#endif

}  // namespace

const VitxEnvSwitch* vitx_env_table(int* n) {
  if (n) *n = kNumSwitches;
  return kSwitches;
}
int vitx_env_diag_build() { return kDiagBuild; }

const char* vitx_env(const char* name) {
  const VitxEnvSwitch* sw = nullptr;
  for (int i = 0; i < kNumSwitches; ++i)
    if (std::strcmp(kSwitches[i].name, name) == 0) { sw = &kSwitches[i]; break; }
  if (!sw) {
    fprintf(stderr, "[vitx] internal error: environment switch %s is not in the registry (csrc/env.hip)\n", name);
    abort();
  }
  const char* v = getenv(name);
  if (v && sw->cls == VITX_ENV_DIAG && !kDiagBuild) {
    static std::mutex mu;
    static std::set<std::string> told;
    std::lock_guard<std::mutex> lk(mu);
    if (told.insert(name).second)
      fprintf(stderr, "[vitx] %s is a diagnostic switch that may corrupt results: ignored by this (release) library; use lib/libvitx_diag.so "
                      "(python vit-tensorflow_amd/build.py --diag)\n", name);
    return nullptr;
  }
  return v;
}
