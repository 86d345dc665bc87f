#!/bin/bash
# round 4, call T: split-operand attention (attn_x3.hip) with the key mask as a template flag and chain-interleaved MFMAs -- parity + same-box A/B
OUT=gpurun_out/r4t; mkdir -p $OUT; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_ref_fixtures.py tests/test_gpu_full_size.py -m gpu -q -k "x3 or bf16x3" --timeout 900 -p no:cacheprovider > $OUT/pytest_x3.log 2>&1; tail -3 $OUT/pytest_x3.log
timeout 1200 python tools/ab_bench.py vit-tensorflow_amd/lib/libvitx_r4p.so vit-tensorflow_amd/lib/libvitx.so 2 -- --compute bf16x3 --batch 256 --steps 6 --warmup 2 > $OUT/ab_x3_attention.log 2>&1; grep -A4 "attn_x3\|\"step\"" $OUT/ab_x3_attention.log | tail -24; grep "round" $OUT/ab_x3_attention.log
