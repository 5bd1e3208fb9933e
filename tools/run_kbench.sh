cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
$K --shape dec360 --dist U --op both --variants 0 --inner 24 --reps 9 --check
$K --shape dec360 --dist M --op both --variants 0 --inner 24 --reps 9 --check
$K --shape dec360 --dist U --B 10 --op both --variants 0 --inner 12 --reps 9
$K --shape dec720 --dist U --op both --variants 0 --inner 8 --reps 7 --check
