#!/bin/bash
# round 4, call U: batched attention GEMM staging (coalesced / conflict-free lane mappings) -- parity, then same-box A/B on DeepViT cfg4 and CaiT cfg5
OUT=gpurun_out/r4u; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deepvit_fused.py tests/test_gpu_ref_fixtures.py tests/test_gpu_full_size_variants.py tests/test_gpu_edges.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for m in deepvit_256 cait_256; do
  timeout 900 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_r4t.so vit-tensorflow_amd/lib/libvitx.so 2 -- --workload $m > $OUT/ab_$m.log 2>&1; echo "== $m"; grep -A4 "attn_bgemm\|\"step\"" $OUT/ab_$m.log | tail -12
done
