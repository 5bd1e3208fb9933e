# round 3, call 54: nt loads of the records only / of the tags only in the record-fed grad_value kernel (A/B builds)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product ntrecs nttags; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape dec360 --dist U --op bwd --variants 0 --check
  timeout 120 $K --shape dec360 --dist M --op bwd --variants 0
  timeout 120 $K --shape dec720 --dist U --op bwd --variants 0 --inner 8
  timeout 120 $K --shape dec360 --dist U --B 10 --op bwd --variants 0 --inner 12
done
} > gpurun_out/c54_kbench.log 2>&1
grep -v "^shape" gpurun_out/c54_kbench.log
