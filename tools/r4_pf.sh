#!/bin/bash
# round 4: one partial-buffer slot per use for the small reductions (no waits for them on the chain) -- side-stream bit-identity + A/B
OUT=gpurun_out/r4pg; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_side_stream.py tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_full_size_variants.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 900 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_head.so vit-tensorflow_amd/lib/libvitx.so 4 > $OUT/ab.log 2>&1; grep "round" $OUT/ab.log; grep -A3 "\"step\"" $OUT/ab.log
for w in cait_256; do timeout 900 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_head.so vit-tensorflow_amd/lib/libvitx.so 2 -- --workload $w > $OUT/ab_$w.log 2>&1; echo "== $w"; grep "round" $OUT/ab_$w.log; done
