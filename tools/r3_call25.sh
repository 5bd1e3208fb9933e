# round 3, call 25: 256-row units / 4 per CU in the tile-fed grad_value kernel (product build)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
( timeout 1200 python -m pytest tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_msda_fused.py tests/test_parity_gaps.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/c25_pytest.log
{
timeout 120 $K --shape enc360 --dist M --op bwd --variants 0,430 --check --inner 8
timeout 120 $K --shape enc360 --dist U --op bwd --variants 0 --inner 8
timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --check --inner 4 --reps 7
timeout 120 $K --shape enc720 --dist M --op bwd --variants 0 --inner 4 --reps 5
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --check --inner 8
timeout 120 $K --shape enc720 --dtype bf16 --B 2 --dist M --op bwd --variants 0 --inner 4 --reps 7
timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --check --inner 8
timeout 120 $K --shape dec360 --dist U --op bwd --variants 0 --check
} > gpurun_out/c25_kbench.log 2>&1
tail -3 gpurun_out/c25_pytest.log; cat gpurun_out/c25_kbench.log
