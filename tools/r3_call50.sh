# round 3, call 50: buffer loads (32-bit offsets) in the tile-fed grad_value kernel's prefetch
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
( timeout 1200 python -m pytest tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_msda_fused.py -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -3 ) > gpurun_out/c50_pytest.log
{
timeout 120 $K --shape enc360 --dist M --op bwd --variants 0 --check --inner 8
timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --check --inner 4 --reps 7
timeout 120 $K --shape enc720 --dist M --op bwd --variants 0 --inner 4 --reps 5
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --check --inner 8
timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --check --inner 8
} > gpurun_out/c50_kbench.log 2>&1
cat gpurun_out/c50_pytest.log; grep -v "^shape" gpurun_out/c50_kbench.log
