#!/bin/bash
# round 4: BF16X3 GEMM kernel compiled with the max-memory-clause scheduling strategy -- BF16X3 parity tests, throughput at batch 64 / 256
OUT=gpurun_out/r4pn; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_ref_fixtures.py tests/test_gpu_full_size.py tests/test_gpu_edges.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
for b in 64 256; do timeout 600 python bench.py --compute bf16x3 --batch $b --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_bf16x3_b$b.json 2> $OUT/bench_bf16x3_b$b.err; python -c "
import json; d=json.load(open('$OUT/bench_bf16x3_b$b.json')); print('bf16x3 b$b', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
