# round 3, call 56: cache policy of the fp32 row gathers (forward and grad_loc kernel): default / sc0 / nt / sc1 (A/B builds)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product tap1 tap2 tap16; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape dec360 --dist U --op both --variants 0 --check
  timeout 120 $K --shape dec360 --dist U --B 10 --op fwd --variants 0 --inner 12
  timeout 120 $K --shape enc360 --dist M --op fwd --variants 0 --inner 8
done
} > gpurun_out/c56_kbench.log 2>&1
grep -v "^shape" gpurun_out/c56_kbench.log
