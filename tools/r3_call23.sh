# round 3, call 23: fused backward through the tile-fed grad_value kernel; 16-bit row stores A/B; B = 10 forward configs
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
( timeout 1200 python -m pytest tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_msda_fused.py tests/test_msda_gpu.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/c23_pytest.log
{
echo "== fused backward: records (430) vs tiles (0)"
timeout 120 $K --shape enc360 --dist M --op fbwd --variants 430,0 --check --inner 8
timeout 120 $K --shape enc360 --dist M --op bwd --variants 0 --inner 8
timeout 120 $K --shape enc720 --dist M --B 2 --op fbwd --variants 430,0 --inner 4 --reps 7
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op fbwd --variants 430,0 --inner 8
timeout 120 $K --shape dec360 --dist U --op fbwd --variants 0 --check
echo "== 16-bit grad_value row stores: nt (product) vs plain (A/B build)"
timeout 120 $K --shape dec720 --dtype bf16 --dist U --op bwd --variants 0 --inner 8
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/plain16 timeout 120 $K --shape dec720 --dtype bf16 --dist U --op bwd --variants 0 --inner 8
timeout 120 $K --shape dec720 --dist U --op bwd --variants 0 --inner 8
timeout 120 $K --shape dec360 --dtype bf16 --dist U --op bwd --variants 0
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/plain16 timeout 120 $K --shape dec360 --dtype bf16 --dist U --op bwd --variants 0
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/plain16 timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --inner 8
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --inner 8
echo "== B = 10 forward configurations"
timeout 120 $K --shape dec360 --dist U --B 10 --op fwd --variants 0,2,3,4,5,12,13,14,15 --inner 12
timeout 120 $K --shape dec360 --dist U --B 20 --op fwd --variants 0,2,3,4,5 --inner 6
} > gpurun_out/c23_kbench.log 2>&1
# PMC: where do the bytes of the 16-bit decoder-720p backward go
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_c23; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT -o bf16_$c -- $GRAFT_REPO_ROOT/tools/kbench.bin --shape dec720 --dtype bf16 --dist U --op bwd --variants 0 --cold-only --inner 2 --reps 2 > /dev/null 2> $OUT/bf16_$c.err
done
rm -f $OUT/*.db $OUT/*agent_info.csv
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/c23_pytest.log; tail -40 gpurun_out/c23_kbench.log
