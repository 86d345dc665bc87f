#!/bin/bash
# round 4, last call: what the driver runs at round end, on the final tree -- GPU test tier, smoke(), the default bench command
OUT=gpurun_out/r4final; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1800 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-330 $OUT/bench_default.json
