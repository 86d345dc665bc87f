#!/bin/bash
# round 4: weight-gradient kernel compiled with the max-ILP scheduling strategy -- full GPU tier + A/B (5 rounds)
OUT=gpurun_out/r4pm; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_all.log 2>&1; grep -E "passed|failed" $OUT/pytest_all.log | tail -2
timeout 900 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_head.so vit-tensorflow_amd/lib/libvitx.so 5 > $OUT/ab.log 2>&1; grep "round" $OUT/ab.log; grep -A3 "\"step\"" $OUT/ab.log
