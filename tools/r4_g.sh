#!/bin/bash
# round 4, call G: 4-wave LayerNorm VJP blocks beside the side-stream weight gradients (A/B), x3 forward with two query blocks per wave
OUT=gpurun_out/r4g; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
echo "=== pytest $(date +%T)"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_side_stream.py tests/test_gpu_ref_fixtures.py tests/test_gpu_edges.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
echo "=== ab LN small blocks $(date +%T)"
timeout 900 python tools/ab_env.py "VITX_LNB_SMALL=0" "VITX_LNB_SMALL=1" "VITX_LNB_SMALL=1 VITX_SIDE_STREAM=0" "VITX_LNB_SMALL=0 VITX_SIDE_STREAM=0" --rounds 3 > $OUT/ab_lnb.log 2>&1; tail -32 $OUT/ab_lnb.log
echo "=== trace $(date +%T)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o side -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $OLDPWD/$OUT/rocprof_bench.json 2> $OLDPWD/$OUT/rocprof.err)
find $OUT/prof -type f ! -name "*.csv" -delete 2>/dev/null
echo "=== x3 b256 $(date +%T)"
timeout 600 python bench.py --compute bf16x3 --batch 256 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_x3_b256.json 2> $OUT/bench_x3.err; python - <<PY
import json
d=json.load(open("$OUT/bench_x3_b256.json"))
print(d["value"], d["ms_per_step"])
for k,v in sorted(d.get("kernel_classes",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:6]: print("   ", k, v["ms_per_step"], v.get("tflops"))
PY
echo "=== done $(date +%T)"
