#!/bin/bash
# one GPU round for the BF16X3 mode: fixture parity tests + the batch-64 bench line (the fp32 parity mode's reference configuration)
TAG=${1:-x3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ref_fixtures.py -q -p no:cacheprovider -k "bf16x3" -s 2>&1 | grep -v "^\[W\|amdgpu.ids" > $OUT/pytest_bf16x3.log
tail -1 $OUT/pytest_bf16x3.log
grep "\[ref" $OUT/pytest_bf16x3.log | awk '{print $1, $4, $9}' | tr "\n" ";"; echo
timeout 300 python bench.py --compute bf16x3 --batch ${2:-64} --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_bf16x3.json 2> $OUT/bench_bf16x3.err
python3 -c "
import json; d=json.load(open('$OUT/bench_bf16x3.json')); print('bf16x3', d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), {k:v['ms_per_step'] for k,v in d['kernel_classes'].items() if v['ms_per_step']>0.5})"
