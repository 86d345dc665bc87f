#!/bin/bash
# round 4, call B: wave-private epilogue -- per-shape sweep against the barrier-round epilogue, correctness, step time
OUT=gpurun_out/r4b; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
echo "=== sweep $(date +%T)"
VITX_SWEEP_VARIANTS=9,12,14,13,15,7 timeout 600 python tools/gemm_sweep.py vitb 20 > $OUT/gemm_sweep_wp.log 2>&1; cat $OUT/gemm_sweep_wp.log | tail -80
echo "=== pytest gemm $(date +%T)"
timeout 1200 python -m pytest tests/test_gpu_parity.py::test_bf16_mfma_gemm_equals_fp32_fma_gemm tests/test_gpu_full_size.py tests/test_gpu_side_stream.py tests/test_gpu_full_size_variants.py -m gpu -q -rA --timeout 900 -p no:cacheprovider > $OUT/pytest_b.log 2>&1; grep -E "^\[|passed|failed|Error|error|^FAILED|^PASSED" $OUT/pytest_b.log | tail -70
echo "=== bench $(date +%T)"
VITX_GEMM_AUTOTUNE_LOG=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json; grep autotune $OUT/bench.err | head -40
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4b/bench.json"))
print(d["ms_per_step"], d.get("roofline",{}).get("frac"))
for r in d.get("gemm_shapes",[]): print(r)
for k,v in d.get("kernel_classes",{}).items(): print(k, v)
PY
echo "=== done $(date +%T)"
