# round 3, call 29: units_min of the tile-fed path (A/B build um1) x queries per tile (variants 0 = 16, 4 = 8, 12 = 8, 5 = 4)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product um1; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0,4,12,5,3 --check --inner 8
  timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0,4 --check --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --op bwd --variants 0,4 --inner 4 --reps 5
done
} > gpurun_out/c29_kbench.log 2>&1
grep -v "^shape" gpurun_out/c29_kbench.log
