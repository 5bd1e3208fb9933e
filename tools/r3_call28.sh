# round 3, call 28: cost of the workgroups past the real unit count (bound forced to the exact 42 / to 300, encoder-360p only)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product bound42 bound300; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0 --check --inner 8
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0 --check --inner 8
done
} > gpurun_out/c28_kbench.log 2>&1
grep -v "^shape" gpurun_out/c28_kbench.log
