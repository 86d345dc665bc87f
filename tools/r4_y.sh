#!/bin/bash
# round 4, call Y: row threshold of the side-stream LayerNorm reductions on CaiT cfg5 (rows of the class-attention layers are 256)
OUT=gpurun_out/r4y; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 900 python tools/ab_env.py "VITX_LN_REDUCE_SIDE_ROWS=1000000000" "VITX_LN_REDUCE_SIDE_ROWS=8192" "VITX_LN_REDUCE_SIDE_ROWS=0" --rounds 3 -- --workload cait_256 > $OUT/ab_env_cait.log 2>&1; tail -12 $OUT/ab_env_cait.log
