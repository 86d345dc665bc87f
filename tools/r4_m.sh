#!/bin/bash
# round 4, call M: DP mode with the pipelined kernel one tile per workgroup (forced DP on one GPU, A/B against the lockstep forms), full GPU tier
OUT=gpurun_out/r4m; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
echo "=== forced DP $(date +%T)"
for k in 0 2; do VITX_FORCE_DP=1 VITX_GEMM_KERNEL=$k VITX_GEMM_AUTOTUNE_LOG=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2> $OUT/dp_k$k.err | cut -c1-260; grep autotune $OUT/dp_k$k.err | awk '{printf "%s ", $16}'; echo; done
echo "=== pytest all $(date +%T)"
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_all.log 2>&1; tail -4 $OUT/pytest_all.log
echo "=== done $(date +%T)"
