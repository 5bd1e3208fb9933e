cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
$K --shape enc360 --dist M --op fwd --variants 0,2,3,4,5,12,13,0 --inner 8 --reps 7
$K --shape enc720 --dist M --B 2 --op fwd --variants 0,2,3,4 --inner 4 --reps 5
