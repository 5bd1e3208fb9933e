cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
$K --shape enc720 --dist M --B 2 --op bwd --variants 0 --inner 4 --reps 5 --check
$K --shape enc360 --dist M --B 2 --op bwd --variants 0 --inner 8 --reps 5 --check
$K --shape enc360 --dist M --B 5 --op bwd --variants 0 --inner 8 --reps 5 --check
$K --shape enc360 --dist U --B 3 --op bwd --variants 0 --inner 8 --reps 5 --check
