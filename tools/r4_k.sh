#!/bin/bash
# round 4, call K: GELU by LDS table in the fc1 epilogue -- sweep A/B, correctness, step A/B
OUT=gpurun_out/r4k; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
for t in 0 1; do echo "=== sweep VITX_GELU_TABLE=$t"; VITX_GELU_TABLE=$t VITX_SWEEP_VARIANTS=13 timeout 300 python tools/gemm_sweep.py vitb 20 2>&1 | grep "epi 2" | sed 's/xp 0: *//'; done | tee $OUT/gemm_sweep_gelu_table.log
echo "=== pytest $(date +%T)"
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_ref_fixtures.py tests/test_gpu_full_size_variants.py -m gpu -q -x --timeout 900 -p no:cacheprovider -rA > $OUT/pytest.log 2>&1; grep -E "passed|failed|^FAILED|^E  |\[gate\] (full_size|gemm_full_size_epi2|bf16)" $OUT/pytest.log | tail -30
echo "=== ab $(date +%T)"
timeout 900 python tools/ab_env.py "VITX_GELU_TABLE=0" "VITX_GELU_TABLE=1" --rounds 3 > $OUT/ab_gelu_table.log 2>&1; tail -16 $OUT/ab_gelu_table.log
echo "=== done $(date +%T)"
