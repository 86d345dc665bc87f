#!/bin/bash
# round 4, call N: BF16X3 with pre-split Dense kernels -- parity + A/B
OUT=gpurun_out/r4n; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_ref_fixtures.py tests/test_gpu_full_size.py tests/test_gpu_wrappers.py tests/test_gpu_distill.py tests/test_gpu_edges.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_x3.log 2>&1; tail -3 $OUT/pytest_x3.log
for b in 64 256; do for ps in 0 1; do VITX_X3_PRESPLIT=$ps timeout 600 python bench.py --compute bf16x3 --batch $b --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_x3_b${b}_ps$ps.json 2> $OUT/bench_x3_b${b}_ps$ps.err; python - <<PY
import json
d=json.load(open("$OUT/bench_x3_b${b}_ps$ps.json"))
print("batch", $b, "presplit", $ps, d["value"], d["ms_per_step"], [ (k, v["ms_per_step"], v.get("tflops")) for k,v in sorted(d.get("kernel_classes",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:2]])
PY
done; done
