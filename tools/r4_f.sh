#!/bin/bash
# round 4, call F: BF16X3 mode with the fused split-operand attention: parity (fixtures, fused vs materialised, full depth), bench b 64 / 256 A/B
OUT=gpurun_out/r4f; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
echo "=== pytest x3 $(date +%T)"
timeout 1500 python -m pytest tests/test_gpu_ref_fixtures.py tests/test_gpu_full_size.py -m gpu -q -rA -k "x3 or bf16x3" --timeout 900 -p no:cacheprovider > $OUT/pytest_x3.log 2>&1; grep -E "^\[|passed|failed|Error|^FAILED|^PASSED" $OUT/pytest_x3.log | tail -40
echo "=== bench x3 $(date +%T)"
for b in 64 256; do for a in 2 1; do echo "batch $b VITX_X3_ATTN=$a"; VITX_X3_ATTN=$a timeout 600 python bench.py --compute bf16x3 --batch $b --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_x3_b${b}_attn$a.json 2> $OUT/bench_x3_b${b}_attn$a.err; python - <<PY
import json
d=json.load(open("$OUT/bench_x3_b${b}_attn$a.json"))
print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))
for k,v in sorted(d.get("kernel_classes",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:9]: print("   ", k, v["ms_per_step"], v.get("tflops"), v.get("gbps"))
PY
done; done
echo "=== done $(date +%T)"
