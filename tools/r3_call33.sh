# round 3, call 33: 16-bit grad_value rows -- head pairs per XCD + plain stores (A/B build pair16)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in product pair16nt; do
  echo "==== $lib"
  if [ $lib = product ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape dec720 --dtype bf16 --dist U --op bwd --variants 0 --check --inner 8
  timeout 120 $K --shape dec360 --dtype bf16 --dist U --op bwd --variants 0 --check
  timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --check --inner 8
  timeout 120 $K --shape enc720 --dtype bf16 --B 2 --dist M --op bwd --variants 0 --inner 4 --reps 7
done
} > gpurun_out/c33_kbench.log 2>&1
grep -v "^shape" gpurun_out/c33_kbench.log
