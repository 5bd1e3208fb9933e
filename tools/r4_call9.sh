cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in base sel4 sel1 sel2 sel3; do
  echo "==== $lib"
  if [ $lib = base ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape dec360 --dist U --op bwd --variants 0,100 --inner 24 --reps 11 --cold-only
  timeout 120 $K --shape dec360 --dist U --B 10 --op bwd --variants 0,100 --inner 12 --reps 11 --cold-only
done
} > gpurun_out/r4c9_kbench.log 2>&1
grep -v "^shape" gpurun_out/r4c9_kbench.log
