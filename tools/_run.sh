mkdir -p gpurun_out/r2e; O=gpurun_out/r2e
python -m pytest tests -m gpu -q -x -rA --timeout 900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
