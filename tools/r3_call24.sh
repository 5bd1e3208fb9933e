# round 3, call 24: tile-fed grad_value kernel with 256-row units, 3 vs 4 units per CU (A/B builds under tools/ab/)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product r256u3 r256u4; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0 --check --inner 8
  timeout 120 $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --inner 4 --reps 7
  timeout 120 $K --shape enc720 --dist M --op bwd --variants 0 --inner 4 --reps 5
  timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --inner 8
  timeout 120 $K --shape enc360 --dist M --op fbwd --variants 0 --inner 8
done
} > gpurun_out/c24_kbench.log 2>&1
cat gpurun_out/c24_kbench.log
