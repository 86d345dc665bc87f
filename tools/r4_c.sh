#!/bin/bash
# round 4, call C: 320-row wave-private variant in the sweep, whole GPU tier, bench
OUT=gpurun_out/r4c; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
echo "=== sweep $(date +%T)"
VITX_SWEEP_VARIANTS=13,11,15,5 timeout 600 python tools/gemm_sweep.py vitb 20 > $OUT/gemm_sweep_wp320.log 2>&1; sed 's/xp 0: *//' $OUT/gemm_sweep_wp320.log | tail -50
echo "=== pytest all $(date +%T)"
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_all.log 2>&1; tail -15 $OUT/pytest_all.log
echo "=== bench $(date +%T)"
VITX_GEMM_AUTOTUNE_LOG=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; grep autotune $OUT/bench.err | head -40
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c/bench.json"))
print(d["ms_per_step"], d.get("roofline",{}).get("frac"))
for r in d.get("gemm_shapes",[]): print(r["form"], r["N"], r["K"], r["epilogue"], r["us"], r["frac"])
for k,v in d.get("kernel_classes",{}).items(): print(k, v["ms_per_step"])
PY
echo "=== done $(date +%T)"
