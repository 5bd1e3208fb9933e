cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=./tools/kbench.bin
( timeout 600 python -m pytest tests/test_msda_gvtiles.py tests/test_parity_r3.py tests/test_parity_gaps.py tests/test_msda_gpu.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c3_pytest.log
{
$K --shape enc360 --dist M --op bwd --variants 430,0,3,2 --check --inner 8
$K --shape enc360 --dist U --op bwd --variants 430,0 --inner 8
$K --shape enc720 --dist M --B 2 --op bwd --variants 430,0 --check --inner 4 --reps 7
$K --shape enc720 --dist M --op bwd --variants 430,0 --inner 4 --reps 5
$K --shape dec360 --dist U --op bwd --variants 0 --check
} > gpurun_out/c3_kbench.log 2>&1
tail -3 gpurun_out/c3_pytest.log
