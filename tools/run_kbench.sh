cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
$K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 3 --timeline
$K --shape dec360 --dist U --op fwd --variants 0 --inner 24 --reps 3 --timeline
