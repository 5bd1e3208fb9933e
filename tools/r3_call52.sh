# round 3, call 52: nt loads of the locations in the grad_loc kernel's small-call configuration (A/B build)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product k1nt product k1nt; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape dec360 --dist U --op bwd --variants 0,100 --check
  timeout 120 $K --shape dec360 --dist M --op bwd --variants 0
done
} > gpurun_out/c52_kbench.log 2>&1
grep -v "^shape" gpurun_out/c52_kbench.log
