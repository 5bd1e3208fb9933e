# round 3, call 36: compact copy of locations / weights for the tile-fed grad_value kernel (510 = on, 511 = off, 0 = by size)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
( timeout 1200 python -m pytest tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_msda_fused.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/c36_pytest.log
{
timeout 120 $K --shape enc360 --dist M --op bwd --variants 511,510 --check --inner 8
timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 511,510 --check --inner 4 --reps 7
timeout 120 $K --shape enc720 --dist M --op bwd --variants 511,510,0 --inner 4 --reps 5
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 511,510 --check --inner 8
timeout 120 $K --shape enc720 --dtype bf16 --B 2 --dist M --op bwd --variants 511,510 --inner 4 --reps 7
timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --check --inner 8
timeout 120 $K --shape enc720 --dist M --B 2 --op fbwd --variants 0 --inner 4 --reps 7
} > gpurun_out/c36_kbench.log 2>&1
tail -3 gpurun_out/c36_pytest.log; grep -v "^shape" gpurun_out/c36_kbench.log
