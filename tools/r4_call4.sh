cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_parity_gaps.py tests/test_parity_r4.py tests/test_msda_gpu.py -m gpu -x -q > gpurun_out/r4c4_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c4_pytest.log )
tail -4 gpurun_out/r4c4_pytest.log
K=./tools/kbench.bin
{
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0,100 --inner 8 --reps 9 --check
  timeout 120 $K --shape enc360 --dist M --dtype bf16 --op bwd --variants 0 --inner 8 --reps 9 --check
  timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --inner 8 --reps 9
  timeout 120 $K --shape enc360 --dist U --op bwd --variants 0 --inner 8 --reps 9 --check
  timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0,100 --inner 4 --reps 7 --check
  timeout 120 $K --shape enc720 --dist M --B 2 --dtype bf16 --op bwd --variants 0 --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --op bwd --variants 0,100 --inner 4 --reps 5
  timeout 120 $K --shape enc720 --dist M --B 1 --op bwd --variants 0 --inner 4 --reps 7
} > gpurun_out/r4c4_kbench.log 2>&1
grep -v "^shape" gpurun_out/r4c4_kbench.log
