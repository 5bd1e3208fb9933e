# round 3, call 40: hardware fp32 -> bf16 conversion (v_cvt_pk_bf16_f32)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
( timeout 1200 python -m pytest tests/test_msda_gpu.py tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_parity_gaps.py tests/test_msda_fused.py -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -3 ) > gpurun_out/c40_pytest.log
{
timeout 120 $K --shape dec720 --dtype bf16 --dist U --op both --variants 0 --check --inner 8
timeout 120 $K --shape dec360 --dtype bf16 --dist U --op both --variants 0 --check
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op both --variants 0 --check --inner 8
timeout 120 $K --shape enc720 --dtype bf16 --B 2 --dist M --op both --variants 0 --inner 4 --reps 7
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op fbwd --variants 0 --inner 8
} > gpurun_out/c40_kbench.log 2>&1
cat gpurun_out/c40_pytest.log; grep -v "^shape" gpurun_out/c40_kbench.log
