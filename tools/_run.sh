mkdir -p gpurun_out/r2l; O=gpurun_out/r2l
timeout 900 python -m pytest tests/test_gpu_ref_fixtures.py tests/test_gpu_distill.py tests/test_efficient_t2t.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest.log 2>&1; tail -30 $O/pytest.log
