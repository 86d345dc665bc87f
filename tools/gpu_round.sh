#!/bin/bash
# One gpurun call: diagnostics, GPU test tier, bench and rocprof summaries -> gpurun_out/<tag>/
# usage: bash tools/gpu_round.sh <tag> [stages...]   (default: all)
TAG=${1:-run}; shift
STAGES=${@:-"probe env gemm parity_fp32 parity_bf16 grads bench_quick pytest bench rocprof"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
for s in $STAGES; do
  echo "=== $s $(date +%T)" | tee -a $OUT/summary.log
  case $s in
    probe) timeout 60 tools/probe_tr > $OUT/probe.log 2>&1; tail -4 $OUT/probe.log | tee -a $OUT/summary.log ;;
    pytest) timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -40 $OUT/pytest.log | tee -a $OUT/summary.log ;;
    bench) timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | tee -a $OUT/summary.log; tail -5 $OUT/bench.err ;;
    rocprof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o vitb16 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $OLDPWD/$OUT/rocprof_bench.json 2> $OLDPWD/$OUT/rocprof.err); ls -R $OUT/prof | head -20; find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -r head -40 | tee -a $OUT/summary.log ;;
    *) timeout 900 python tools/gpu_diag.py $s > $OUT/$s.log 2>&1; tail -60 $OUT/$s.log ;;
  esac
done
# keep the merged-back payload small: drop raw traces, keep csv summaries
find $OUT/prof -type f ! -name "*.csv" -delete 2>/dev/null
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
du -sh $OUT
