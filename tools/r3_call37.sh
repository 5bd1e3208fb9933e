# round 3, call 37: fp32 rows of the grad_loc kernel as 4 lanes x 32 B (A/B builds: 1 = large calls, 2 = all)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product k1lpr4_1 k1lpr4_2; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0 --check --inner 8
  timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --check --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --op bwd --variants 0 --inner 4 --reps 5
  timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --check --inner 8
  timeout 120 $K --shape dec360 --dist U --op bwd --variants 0 --check
  timeout 120 $K --shape dec720 --dist U --op bwd --variants 0 --check --inner 8
done
} > gpurun_out/c37_kbench.log 2>&1
grep -v "^shape" gpurun_out/c37_kbench.log
