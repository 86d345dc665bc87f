#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 900 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_head.so vit-tensorflow_amd/lib/libvitx.so 5 > $OUT/ab2.log 2>&1; grep "round" $OUT/ab2.log; grep -A3 "\"step\"" $OUT/ab2.log
