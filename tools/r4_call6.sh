cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in base gvt1 gvt2 gvt3; do
  echo "==== $lib"
  if [ $lib = base ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0,100 --inner 8 --reps 9 --cold-only
  timeout 120 $K --shape enc720 --dist M --op bwd --variants 0,100 --inner 4 --reps 5 --cold-only
done
} > gpurun_out/r4c6_kbench.log 2>&1
grep -v "^shape" gpurun_out/r4c6_kbench.log
