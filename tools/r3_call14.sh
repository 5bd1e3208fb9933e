cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=./tools/kbench.bin
( timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_parity_gaps.py tests/test_parity_r3.py tests/test_msda_gvtiles.py tests/test_msda_fused.py tests/test_reid.py tests/test_oracle.py -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/c14_pytest.log
{
timeout 120 $K --shape dec360 --dist U --op both --variants 0 --check --timeline
timeout 120 $K --shape dec360 --dist M --op bwd --variants 0 --check
timeout 120 $K --shape dec720 --dist U --op bwd --variants 0 --check --inner 8
timeout 120 $K --shape enc360 --dist M --op bwd --variants 0,430 --check --inner 8
} > gpurun_out/c14_kbench.log 2>&1
tail -3 gpurun_out/c14_pytest.log
