# round 3, call 55: nt loads of tags / records in the record-fed grad_value kernel only when the grid is one round
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
( timeout 1200 python -m pytest tests/test_msda_gpu.py tests/test_parity_gaps.py tests/test_msda_gvtiles.py -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -2 ) > gpurun_out/c55_pytest.log
{
timeout 120 $K --shape dec360 --dist U --op bwd --variants 0 --check
timeout 120 $K --shape dec360 --dist M --op bwd --variants 0
timeout 120 $K --shape dec720 --dist U --op bwd --variants 0 --inner 8
timeout 120 $K --shape dec360 --dist U --B 10 --op bwd --variants 0 --inner 12
timeout 120 $K --shape dec360 --dtype bf16 --dist U --op bwd --variants 0
timeout 120 $K --shape dec360 --dist U --B 2 --op bwd --variants 0
} > gpurun_out/c55_kbench.log 2>&1
cat gpurun_out/c55_pytest.log; grep -v "^shape" gpurun_out/c55_kbench.log
