# round 3, call 45: samples in flight per lane group in the grad_loc kernel's large-call configuration (A/B builds)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product k1b1 k1b4; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0 --check --inner 8
  timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --op bwd --variants 0 --inner 4 --reps 5
  timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --inner 8
done
} > gpurun_out/c45_kbench.log 2>&1
grep -v "^shape" gpurun_out/c45_kbench.log
