# round 4, call 1: full GPU suite on the product / development split, backward PMC, kbench baselines, two A/Bs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4c1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c1_pytest.log )
tail -5 gpurun_out/r4c1_pytest.log
K=./tools/kbench.bin
{
  timeout 120 $K --shape dec360 --dist U --op both --variants 0 --check
  timeout 120 $K --shape dec360 --dist M --op both --variants 0 --check
  timeout 120 $K --shape dec360 --dist U --B 10 --op both --variants 0 --check --inner 12
  timeout 120 $K --shape enc360 --dist M --op both --variants 0 --check --inner 8
  timeout 120 $K --shape enc360 --dist M --dtype bf16 --op both --variants 0 --check --inner 8
  timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --inner 8
  timeout 120 $K --shape dec720 --dist U --op both --variants 0 --inner 8
  timeout 120 $K --shape dec720 --dist U --dtype bf16 --op both --variants 0 --inner 8
  timeout 120 $K --shape enc720 --dist M --B 2 --op both --variants 0 --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --B 2 --dtype bf16 --op both --variants 0 --check --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --op both --variants 0 --inner 4 --reps 5
  echo "==== rounds3 (VNX_TILE_ROUNDS=3: one selection window per level at 360p)"
  export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/rounds3
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0 --check --inner 8
  timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --op bwd --variants 0 --inner 4 --reps 5
  unset LD_LIBRARY_PATH
} > gpurun_out/r4c1_kbench.log 2>&1
grep -v "^shape" gpurun_out/r4c1_kbench.log
bash tools/prof_backward_pmc.sh r04 > gpurun_out/r4c1_pmc.log 2>&1
tail -30 gpurun_out/r4c1_pmc.log
