cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/vnext_amd/lib/expst $K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 5 --timeline
