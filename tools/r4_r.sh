#!/bin/bash
# round 4, call R: bf16 attention kernels (instruction-count rewrite, operand prefetch) -- timeline probe, correctness, same-box A/B against the r4p library
OUT=gpurun_out/r4r; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
echo "=== probe $(date +%T)"; tools/probe_attn 2>&1 | grep -v "start of" | tee $OUT/probe_attn.log
echo "=== pytest $(date +%T)"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_full_size.py tests/test_gpu_ref_fixtures.py tests/test_gpu_full_size_variants.py tests/test_gpu_dropout.py -m gpu -q --timeout 900 -p no:cacheprovider -rA > $OUT/pytest.log 2>&1; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest.log | tail -30
echo "=== ab $(date +%T)"
timeout 900 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_r4p.so vit-tensorflow_amd/lib/libvitx.so 3 > $OUT/ab_attention.log 2>&1; grep -A4 "attn_bf16\|\"step\"" $OUT/ab_attention.log | tail -24
echo "=== done $(date +%T)"
