#!/bin/bash
# round 4, call V: bf16 operand refresh as ONE launch over a table of all Dense kernels -- parity tier, then same-box A/B
OUT=gpurun_out/r4v; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_all.log 2>&1; tail -3 $OUT/pytest_all.log
timeout 900 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_r4t.so vit-tensorflow_amd/lib/libvitx.so 3 > $OUT/ab_convert.log 2>&1; grep -A4 "convert_weights\|\"step\"" $OUT/ab_convert.log | tail -12
