mkdir -p gpurun_out/r2j; O=gpurun_out/r2j
python -m pytest tests -m gpu -q -x -rA --timeout 900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log; grep -E "FAILED|Error|error" $O/pytest_gpu.log | head -10
VITX_F32_MFMA=0 python bench.py --compute fp32 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fp32_scalar.json 2>$O/bench_fp32_scalar.err; cut -c1-300 $O/bench_fp32_scalar.json
python bench.py --compute fp32 --batch 64 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_fp32_mfma.json 2>$O/bench_fp32_mfma.err; cat $O/bench_fp32_mfma.json
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-3000
