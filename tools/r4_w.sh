#!/bin/bash
# round 4, call W2: evidence for the final tree -- smoke(), the driver's default bench command, first-step cost, rocprofv3 kernel stats (one stream),
# the other BASELINE workloads, the BF16X3 and fp32 modes
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
OUT=gpurun_out/r4w2; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== smoke $(date +%T)"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $OUT/smoke.log
echo "=== default bench $(date +%T)"; timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])"
VITX_SIDE_STREAM=0 bash tools/gpu_round.sh r4w2 rocprof > $OUT/rocprof_stage.log 2>&1; tail -3 $OUT/rocprof_stage.log
for w in vit_l16_224 deepvit_256 cait_256 vit_b16_256 vit_readme_256; do timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err; python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('path_mfma_frac'))"; done
for b in 64 256; do timeout 600 python bench.py --compute bf16x3 --batch $b --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_bf16x3_b$b.json 2> $OUT/bench_bf16x3_b$b.err; python -c "
import json; d=json.load(open('$OUT/bench_bf16x3_b$b.json')); print('bf16x3 b$b', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
timeout 600 python bench.py --compute fp32 --batch 64 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_fp32_b64.json 2> $OUT/bench_fp32_b64.err; python -c "
import json; d=json.load(open('$OUT/bench_fp32_b64.json')); print('fp32 b64', d['value'], d['ms_per_step'])"
find $OUT -name "*kernel_trace.csv" -size +5M -delete 2>/dev/null
du -sh $OUT
