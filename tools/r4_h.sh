#!/bin/bash
# round 4, call H: evidence for the tree as it stands -- driver-style bench line, rocprofv3 kernel stats, PMC traffic, the other BASELINE workloads
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
bash tools/gpu_round.sh r4h bench
VITX_SIDE_STREAM=0 bash tools/gpu_round.sh r4h rocprof pmc_bench
OUT=gpurun_out/r4h
python tools/rocprof_family_summary.py $OUT/prof > $OUT/rocprofv3_family_summary.json 2> $OUT/family.err; head -c 1200 $OUT/rocprofv3_family_summary.json
for w in vit_l16_224 deepvit_256 cait_256 vit_b16_256 vit_readme_256; do timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err; python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('path_mfma_frac'))"; done
find $OUT -name "*kernel_trace.csv" -size +5M -delete 2>/dev/null
du -sh $OUT
