cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=./tools/kbench.bin
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c1_pytest.log
{
$K --shape dec360 --dist U --op both --variants 0 --check
$K --shape dec360 --dist U --op fwd --variants 0,13,33,53 --timeline
$K --shape dec360 --dist M --op fwd --variants 0,33
$K --shape dec360 --dist U --B 10 --op fwd --variants 0,33
$K --shape enc360 --dist M --op both --variants 0 --check --inner 8
$K --shape dec720 --dist U --op both --variants 0 --inner 8
$K --shape enc720 --dist M --op both --variants 0 --inner 4 --reps 5
} > gpurun_out/c1_kbench.log 2>&1
tail -5 gpurun_out/c1_pytest.log
