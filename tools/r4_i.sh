#!/bin/bash
# round 4, call I: MPP wrapper on the GPU (fixture + oracle), the other wrappers still green
OUT=gpurun_out/r4i; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_wrappers.py tests/test_gpu_distill.py -m gpu -q -rA --timeout 900 -p no:cacheprovider > $OUT/pytest_wrappers.log 2>&1; grep -E "^\[gate|passed|failed|Error|^FAILED|^PASSED.*mpp|^E " $OUT/pytest_wrappers.log | tail -40
