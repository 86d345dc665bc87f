#!/bin/bash
# round 4, call J: whole GPU tier on the tree with MPP + fused x3 attention (two query blocks per wave)
OUT=gpurun_out/r4j; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 2400 python -m pytest tests -m gpu -q -rA --timeout 900 -p no:cacheprovider > $OUT/pytest_all.log 2>&1; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_all.log | tail -20
