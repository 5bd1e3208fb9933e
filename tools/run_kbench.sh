cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
$K --shape dec360 --dist U --op bwd --variants 0,0 --inner 24 --reps 9 --check
$K --shape enc360 --dist M --op bwd --variants 0 --inner 8 --reps 7 --check
$K --shape dec720 --dist U --op bwd --variants 0 --inner 8 --reps 7 --check
