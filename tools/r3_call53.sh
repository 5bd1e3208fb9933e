# round 3, call 53: nt loads of tags / records in the record-fed grad_value kernel, of grad_out rows in the grad_loc kernel (A/B)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product selnt gont product selnt gont; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape dec360 --dist U --op bwd --variants 0 --check
  timeout 120 $K --shape dec360 --dist M --op bwd --variants 0
done
} > gpurun_out/c53_kbench.log 2>&1
grep -v "^shape" gpurun_out/c53_kbench.log
