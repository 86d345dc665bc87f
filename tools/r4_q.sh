#!/bin/bash
# round 4, call Q: the parity tier with one GEMM variant forced everywhere (ragged tiles of every test shape through the generic wave-private form),
# and with the A/B switches off
OUT=gpurun_out/r4q; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
T="tests/test_gpu_parity.py tests/test_gpu_ref_fixtures.py tests/test_gpu_wrappers.py tests/test_gpu_distill.py tests/test_gpu_edges.py tests/test_gpu_dropout.py tests/test_gpu_deepvit_fused.py"
for k in 13 11; do echo "=== VITX_GEMM_KERNEL=$k"; VITX_GEMM_KERNEL=$k timeout 1200 python -m pytest $T -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_kernel$k.log 2>&1; grep -E "passed|failed|^FAILED" $OUT/pytest_kernel$k.log | tail -8; done
echo "=== switches off"; VITX_SIDE_STREAM=0 VITX_GELU_TABLE=0 timeout 1200 python -m pytest $T tests/test_gpu_full_size.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_switches_off.log 2>&1; grep -E "passed|failed|^FAILED" $OUT/pytest_switches_off.log | tail -5
