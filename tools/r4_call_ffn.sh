cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_fused_ffn.py tests/test_fused_norm.py tests/test_transformer.py tests/test_idol_model.py tests/test_model_ddp.py -m gpu -x -q > gpurun_out/r4_ffn_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4_ffn_pytest.log )
tail -25 gpurun_out/r4_ffn_pytest.log | cut -c1-300
export PYTHONPATH=$GRAFT_REPO_ROOT

