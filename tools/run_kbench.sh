cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
$K --dma-test --shape dec360 --dist U --op both --variants 0 --check
$K --shape dec360 --dist M --op both --variants 0 --check
$K --shape dec360 --dist U --B 10 --op fwd --variants 0
$K --shape enc360 --dist M --op both --variants 0 --check --inner 8
$K --shape enc360 --dist U --op both --variants 0 --inner 8
$K --shape dec720 --dist U --op both --variants 0 --inner 8
$K --shape enc720 --dist M --op both --variants 0 --inner 4 --reps 5
