cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
for lib in lib lib/exk1nt lib lib/exk1nt; do
echo "== $lib"
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/vnext_amd/$lib $K --shape dec360 --dist U --op bwd --variants 0 --inner 24 --reps 9
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/vnext_amd/$lib $K --shape enc360 --dist M --op bwd --variants 0,100 --inner 8 --reps 7
done
