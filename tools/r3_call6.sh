cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=./tools/kbench.bin
( timeout 600 python -m pytest tests/test_msda_tile.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/c6_pytest.log
{
timeout 120 $K --shape enc360 --dist M --op fwd --variants 0,700,720 --check --inner 8
timeout 120 $K --shape enc720 --dist M --B 2 --op fwd --variants 0,700,720 --check --inner 4 --reps 7
timeout 120 $K --shape enc720 --dist M --op fwd --variants 0,720 --inner 4 --reps 5
} > gpurun_out/c6_kbench.log 2>&1
tail -3 gpurun_out/c6_pytest.log
