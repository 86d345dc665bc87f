#!/bin/bash
# round 4, call O: finer K slices of the weight-gradient GEMMs beside the side stream (A/B), same box
OUT=gpurun_out/r4o; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 900 python tools/ab_env.py "VITX_TN_WORKGROUPS=256" "VITX_TN_WORKGROUPS=512" "VITX_TN_WORKGROUPS=384" "VITX_TN_WORKGROUPS=512 VITX_SIDE_STREAM=0" --rounds 3 > $OUT/ab_tn_slices.log 2>&1; tail -30 $OUT/ab_tn_slices.log
