#!/bin/bash
# round 4, call E: cleaned-up GEMM family (wave-private epilogue only, 256 / 320 rows): sweep, whole GPU tier, A/B against the round-3 library
OUT=gpurun_out/r4e; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
echo "=== sweep $(date +%T)"
VITX_SWEEP_VARIANTS=13,11,5 timeout 600 python tools/gemm_sweep.py vitb 20 > $OUT/gemm_sweep.log 2>&1; sed 's/xp 0: *//' $OUT/gemm_sweep.log | tail -40
echo "=== pytest all $(date +%T)"
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_all.log 2>&1; tail -8 $OUT/pytest_all.log
echo "=== ab r3 vs r4 $(date +%T)"
timeout 1200 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_r3.so vit-tensorflow_amd/lib/libvitx.so 3 > $OUT/ab_r3_r4.log 2>&1; grep "round" $OUT/ab_r3_r4.log; grep -A4 '"gemm_bf16_mfma"\|"step"' $OUT/ab_r3_r4.log
echo "=== first step cost + autotune picks $(date +%T)"
for i in 1 2 3; do VITX_GEMM_AUTOTUNE_LOG=1 timeout 300 python bench.py --steps 5 --warmup 0 --no-cpu-baseline --no-profile 2> $OUT/picks_$i.err | cut -c1-200; grep autotune $OUT/picks_$i.err | awk '{print $5,$7,$9,$11,$15}' | tr '\n' ';'; echo; done
echo "=== done $(date +%T)"
