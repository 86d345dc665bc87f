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
    counters) (cd /tmp && rocprofv3 -L > $OLDPWD/$OUT/counters_list.txt 2>&1); grep -c SQ_ $OUT/counters_list.txt ;;
    pmc_gemm) for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
        tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
        (cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc_$tag -o g -- python $OLDPWD/tools/gemm_bench.py ${PMC_SHAPE:-8192 8192 8192 2 3 3} > $OLDPWD/$OUT/pmc_$tag.log 2>&1); tail -2 $OUT/pmc_$tag.log
      done ;;
    pmc_sq) for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do   # (GRBM has its own two slots: tools/pmc_sq_summary.py needs GRBM_GUI_ACTIVE in the instruction pass)
        tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
        (cd /tmp && timeout 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OLDPWD/$OUT/sq_$tag -o b -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $OLDPWD/$OUT/sq_$tag.json 2> $OLDPWD/$OUT/sq_$tag.err); tail -c 100 $OUT/sq_$tag.json
      done
      find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
      python tools/pmc_sq_summary.py $OUT/sq_SQ_INSTS_VALU_SQ_INSTS_MFMA_SQ_VALU_MFMA/b_counter_collection.csv $OUT/sq_SQ_WAVE_CYCLES_SQ_BUSY_CYCLES_SQ_WAIT_AN/b_counter_collection.csv > $OUT/pmc_sq_summary.txt 2>&1; head -12 $OUT/pmc_sq_summary.txt ;;
    rocprof1) (cd /tmp && VITX_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof1 -o vitb16 -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/rocprof1_bench.json 2> $OLDPWD/$OUT/rocprof1.err)
      python tools/rocprof_family_summary.py $(find $OUT/prof1 -name "*kernel_stats.csv" | head -1) $(find $OUT/prof1 -name "*kernel_trace.csv" | head -1) 5 > $OUT/rocprofv3_family_summary.json 2>$OUT/rocprof1_summary.err; head -c 1500 $OUT/rocprofv3_family_summary.json
      find $OUT/prof1 -name "*kernel_trace*" -delete 2>/dev/null ;;
    pmc_bench) for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
        tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
        (cd /tmp && timeout 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc_$tag -o b -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $OLDPWD/$OUT/pmc_$tag.json 2> $OLDPWD/$OUT/pmc_$tag.err); tail -c 300 $OUT/pmc_$tag.json
      done
      find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
      python tools/pmc_traffic.py $OUT gemm_bf16_mfma=99,gemm_bf16_mfma_tn=50 > $OUT/pmc_traffic.json 2>$OUT/pmc_traffic.err; head -c 1500 $OUT/pmc_traffic.json ;;
    *) timeout 900 python tools/gpu_diag.py $s > $OUT/$s.log 2>&1; tail -60 $OUT/$s.log ;;
  esac
done
# keep the merged-back payload small: drop raw traces, keep csv summaries
find $OUT/prof -type f ! -name "*.csv" -delete 2>/dev/null
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
du -sh $OUT
