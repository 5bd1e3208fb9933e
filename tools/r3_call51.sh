# round 3, call 51: 24-bit index multiplies / shift instead of a division in the decode phase of the forward and grad_loc kernels
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
( timeout 1200 python -m pytest tests/test_msda_gpu.py tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_parity_gaps.py tests/test_msda_fused.py tests/test_msda_tile.py -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -3 ) > gpurun_out/c51_pytest.log
{
timeout 120 $K --shape dec360 --dist U --op both --variants 0 --check
timeout 120 $K --shape dec360 --dist M --op both --variants 0 --check
timeout 120 $K --shape dec720 --dist U --op both --variants 0 --check --inner 8
timeout 120 $K --shape enc360 --dist M --op both --variants 0 --check --inner 8
timeout 120 $K --shape enc720 --dist M --op both --variants 0 --inner 4 --reps 5
timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --check --inner 8
} > gpurun_out/c51_kbench.log 2>&1
cat gpurun_out/c51_pytest.log; grep -v "^shape" gpurun_out/c51_kbench.log
