cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
( timeout 900 python -m pytest tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_parity_gaps.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/c22_pytest.log
{
timeout 120 $K --shape enc360 --dist M --op bwd --variants 0,430 --check --inner 8
timeout 120 $K --shape enc360 --dist U --op bwd --variants 0 --inner 8
timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --check --inner 4 --reps 7
timeout 120 $K --shape enc720 --dist M --op bwd --variants 0 --inner 4 --reps 5
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --inner 8
} > gpurun_out/c22_kbench.log 2>&1
tail -3 gpurun_out/c22_pytest.log
