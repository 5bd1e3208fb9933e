cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K=./tools/kbench.bin
( timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_parity_r3.py tests/test_parity_gaps.py tests/test_msda_fused.py tests/test_msda_gvtiles.py -m gpu -x -q -k "16 or bf16 or half or sixteen" 2>&1 | tail -15 ) > gpurun_out/c12_pytest.log
{
timeout 120 $K --shape dec720 --dtype bf16 --dist U --op both --variants 69,0 --check --inner 8
timeout 120 $K --shape dec720 --dtype f32 --dist U --op both --variants 0 --inner 8
timeout 120 $K --shape dec360 --dtype bf16 --dist U --op both --variants 69,0 --check
timeout 120 $K --shape enc360 --dtype bf16 --dist M --op both --variants 69,0 --check --inner 8
timeout 120 $K --shape enc720 --dtype bf16 --dist M --B 2 --op both --variants 69,0 --inner 4 --reps 7
timeout 120 $K --shape enc360 --dtype f32 --dist M --op fwd --variants 0,720 --inner 8
} > gpurun_out/c12_kbench.log 2>&1
tail -3 gpurun_out/c12_pytest.log
