cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest "tests/test_parity_r3.py::test_encoder_720p_all_rows_all_gradients_fp32" -x -q 2>&1 | tail -60 ) > gpurun_out/c2_fail.log
( timeout 900 python -m pytest tests -m gpu -q --deselect "tests/test_parity_r3.py::test_encoder_720p_all_rows_all_gradients_fp32" 2>&1 | tail -40 ) > gpurun_out/c2_pytest.log
tail -3 gpurun_out/c2_pytest.log
