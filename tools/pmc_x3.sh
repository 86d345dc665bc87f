#!/bin/bash
# SQ counter passes over the bf16x3 GEMM kernel inside a short bench run (what do its waves wait on?):  bash tools/pmc_x3.sh <tag>
TAG=${1:-pmcx3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OLDPWD/$OUT/p$i -o g -- python $OLDPWD/bench.py --compute bf16x3 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $OLDPWD/$OUT/p$i.log 2>&1)
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i" >> $OUT/summary.txt
  [ -n "$f" ] && python tools/pmc_sum.py $f gemm_bf16x3_kernel >> $OUT/summary.txt
done
find $OUT -name "*.csv" -delete 2>/dev/null
cat $OUT/summary.txt
