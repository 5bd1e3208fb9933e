# round 4, call 2: where do the instructions of the two encoder-shape backward kernels go?  Timing ablations (A/B builds)
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
{
for lib in base gvt1 gvt2 gvt3 k1a1 k1a2 k1a4 k1a8; do
  echo "==== $lib"
  if [ $lib = base ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$lib; fi
  timeout 120 $K --shape enc360 --dist M --op bwd --variants 0,100 --inner 8 --reps 9 --cold-only
  case $lib in base|k1a*) timeout 120 $K --shape dec360 --dist U --op bwd --variants 0,100 --inner 24 --reps 9 --cold-only;; esac
done
} > gpurun_out/r4c2_kbench.log 2>&1
grep -v "^shape" gpurun_out/r4c2_kbench.log
