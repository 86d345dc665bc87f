#!/bin/bash
# round 4, call A: side-stream correctness + A/B, full-size parity of cfg3/4/5, kernel trace with the side stream on
OUT=gpurun_out/r4a; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
echo "=== pytest side stream $(date +%T)"
timeout 900 python -m pytest tests/test_gpu_side_stream.py -m gpu -q -rA --timeout 600 -p no:cacheprovider > $OUT/pytest_side.log 2>&1; tail -15 $OUT/pytest_side.log
echo "=== ab side stream $(date +%T)"
timeout 900 python tools/ab_env.py "VITX_SIDE_STREAM=0" "VITX_SIDE_STREAM=1" "VITX_SIDE_STREAM=2" --rounds 3 > $OUT/ab_side.log 2>&1; tail -30 $OUT/ab_side.log
echo "=== pytest full size variants $(date +%T)"
timeout 1200 python -m pytest tests/test_gpu_full_size_variants.py -m gpu -q -rA --timeout 900 -p no:cacheprovider > $OUT/pytest_fullsize.log 2>&1; grep -E "^\[|passed|failed|Error|error" $OUT/pytest_fullsize.log | tail -40
echo "=== rocprof side stream $(date +%T)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o side -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $OLDPWD/$OUT/rocprof_bench.json 2> $OLDPWD/$OUT/rocprof.err)
cat $OUT/rocprof_bench.json | cut -c1-300
find $OUT/prof -type f ! -name "*.csv" -delete 2>/dev/null
ls -la $OUT/prof/* | head
echo "=== done $(date +%T)"
