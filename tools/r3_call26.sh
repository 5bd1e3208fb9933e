# round 3, call 26: rectangle units (blocks for wide levels) in the tile-fed grad_value kernel
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
( timeout 1200 python -m pytest tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_msda_fused.py tests/test_parity_gaps.py -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/c26_pytest.log
{
timeout 120 $K --shape enc360 --dist M --op bwd --variants 0 --check --inner 8
timeout 120 $K --shape enc360 --dist U --op bwd --variants 0 --inner 8
timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --check --inner 4 --reps 7
timeout 120 $K --shape enc720 --dist M --op bwd --variants 0 --inner 4 --reps 5
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --check --inner 8
timeout 120 $K --shape enc720 --dtype bf16 --B 2 --dist M --op bwd --variants 0 --inner 4 --reps 7
timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --check --inner 8
timeout 120 $K --shape enc720 --dist M --B 2 --op fbwd --variants 0 --inner 4 --reps 7
} > gpurun_out/c26_kbench.log 2>&1
tail -5 gpurun_out/c26_pytest.log; cat gpurun_out/c26_kbench.log
