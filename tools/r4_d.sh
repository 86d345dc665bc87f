#!/bin/bash
# round 4, call D: same-box A/B of the round-3 library against this tree (step + per-class), side stream on/off inside the new one
OUT=gpurun_out/r4d; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
echo "=== ab r3 vs r4 $(date +%T)"
timeout 1200 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_r3.so vit-tensorflow_amd/lib/libvitx.so 3 > $OUT/ab_r3_r4.log 2>&1; tail -60 $OUT/ab_r3_r4.log
echo "=== done $(date +%T)"
