#!/bin/bash
# round 4, call L: x3 attention backward with two blocks per wave -- parity + bench
OUT=gpurun_out/r4l; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_ref_fixtures.py tests/test_gpu_full_size.py -m gpu -q -k "x3 or bf16x3" --timeout 900 -p no:cacheprovider > $OUT/pytest_x3.log 2>&1; tail -3 $OUT/pytest_x3.log
for b in 64 256; do timeout 600 python bench.py --compute bf16x3 --batch $b --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_x3_b$b.json 2> $OUT/bench_x3_b$b.err; python - <<PY
import json
d=json.load(open("$OUT/bench_x3_b$b.json"))
print($b, d["value"], d["ms_per_step"])
for k,v in sorted(d.get("kernel_classes",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:4]: print("   ", k, v["ms_per_step"], v.get("tflops"))
PY
done
