#!/bin/bash
# SQ counter passes over the 8192^3 NT GEMM micro-benchmark for the timing-switch variants (xp 0 / 2 / 4): what do the waves wait on?
# usage: bash tools/pmc_gemm_xp.sh <tag> [variant]
TAG=${1:-pmcxp}; V=${2:-14}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp VITX_GEMM_XP=1
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
for x in 0 2 4 6; do
  K=$((V + 16 * x))
  i=0
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    (cd /tmp && timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OLDPWD/$OUT/x${x}_p$i -o g -- python $OLDPWD/tools/gemm_bench.py 8192 8192 8192 $K 3 3 > $OLDPWD/$OUT/x${x}_p$i.log 2>&1)
    f=$(find $OUT/x${x}_p$i -name "*counter_collection.csv" | head -1)
    echo "== xp $x pass $i: $(tail -1 $OUT/x${x}_p$i.log)" >> $OUT/summary.txt
    [ -n "$f" ] && python tools/pmc_sum.py $f gemm_bf16_nt_pipe >> $OUT/summary.txt
  done
done
find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
find $OUT -name "*.csv" -size +5M -delete 2>/dev/null
cat $OUT/summary.txt
