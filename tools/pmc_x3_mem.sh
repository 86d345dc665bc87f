#!/bin/bash
# L2 / fabric counters of the bf16x3 GEMM kernel inside a short bench run:  bash tools/pmc_x3_mem.sh <tag>
TAG=${1:-pmcx3m}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
i=0
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i + 1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OLDPWD/$OUT/p$i -o g -- python $OLDPWD/bench.py --compute bf16x3 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $OLDPWD/$OUT/p$i.log 2>&1)
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i" >> $OUT/summary.txt
  [ -n "$f" ] && python tools/pmc_sum.py $f gemm_bf16x3_kernel >> $OUT/summary.txt || tail -3 $OUT/p$i.log >> $OUT/summary.txt
done
find $OUT -name "*.csv" -delete 2>/dev/null
cat $OUT/summary.txt
