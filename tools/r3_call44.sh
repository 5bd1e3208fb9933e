# round 3, call 44: query-split pieces at small batches (B x M < 32 doubles VNX_QS_MID): A/B builds
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product mid4 mid8 mid4c16; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --check --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --B 1 --op bwd --variants 0 --inner 4 --reps 7
  timeout 120 $K --shape enc360 --dist M --B 2 --op bwd --variants 0 --inner 8
  timeout 120 $K --shape enc360 --dist M --B 1 --op bwd --variants 0 --inner 8
  timeout 120 $K --shape enc360 --dist M --B 3 --op bwd --variants 0 --inner 8
done
} > gpurun_out/c44_kbench.log 2>&1
grep -v "^shape" gpurun_out/c44_kbench.log
