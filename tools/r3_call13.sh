cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=./tools/kbench.bin
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c13_pytest.log
{
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op bwd --variants 0 --check --inner 8
timeout 120 $K --shape enc720 --dtype bf16 --dist M --B 2 --op bwd --variants 0 --inner 4 --reps 7
timeout 120 $K --shape dec360 --dtype f32 --dist U --op both --variants 0 --check
} > gpurun_out/c13_kbench.log 2>&1
tail -3 gpurun_out/c13_pytest.log
