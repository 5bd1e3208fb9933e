cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4c8_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c8_pytest.log )
tail -3 gpurun_out/r4c8_pytest.log
bash tools/run_kbench.sh > gpurun_out/r4c8_kbench.log 2>&1
K=./tools/kbench.bin
{
  timeout 120 $K --shape enc360 --dist M --dtype bf16 --op both --variants 0 --check --inner 8
  timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --inner 8
  timeout 120 $K --shape dec720 --dist U --dtype bf16 --op both --variants 0 --inner 8
  timeout 120 $K --shape enc720 --dist M --B 2 --op both --variants 0 --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --B 2 --dtype bf16 --op both --variants 0 --check --inner 4 --reps 7
  timeout 120 $K --shape dec360 --dist U --B 10 --op both --variants 0 --inner 12
} >> gpurun_out/r4c8_kbench.log 2>&1
grep -v "^shape" gpurun_out/r4c8_kbench.log
