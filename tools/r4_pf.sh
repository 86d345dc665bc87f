#!/bin/bash
OUT=gpurun_out/r4pi; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 900 python tools/ab_env.py "VITX_SIDE_STREAM=1" "VITX_SIDE_STREAM=2" "VITX_SIDE_STREAM=0" --rounds 3 > $OUT/ab_env_side_mode.log 2>&1; tail -8 $OUT/ab_env_side_mode.log
