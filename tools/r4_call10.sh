cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
  timeout 120 $K --shape enc720 --dist M --op bwd --variants 0,510,100 --inner 4 --reps 5 --cold-only --check
  timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0,510,100 --inner 4 --reps 7 --cold-only
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0,510,100 --inner 8 --reps 9 --cold-only
} > gpurun_out/r4c10_kbench.log 2>&1
grep -v "^shape" gpurun_out/r4c10_kbench.log
