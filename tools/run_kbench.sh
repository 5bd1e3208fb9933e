cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
$K --shape enc360 --dist M --op fwd --variants 710,0 --check --inner 8
$K --shape enc360 --dist U --op fwd --variants 710,0 --check --inner 8
$K --shape enc360 --dist M --op ffwd --variants 710,0 --check --inner 8
$K --shape enc720 --dist M --op ffwd --variants 710,0 --check --inner 4 --reps 5
$K --shape dec360 --dist U --op ffwd --variants 0,700 --check
