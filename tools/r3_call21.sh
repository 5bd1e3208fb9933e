cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_mask_head.py tests/test_fused_norm.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/c21_pytest.log
python tools/prof_heads.py > gpurun_out/c21_heads.json 2> gpurun_out/c21_heads.err
tail -3 gpurun_out/c21_pytest.log
