cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
for lib in lib lib/exk1; do
echo "== $lib"
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/vnext_amd/$lib $K --shape enc360 --dist M --op bwd --variants 0,100 --inner 8 --reps 7
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/vnext_amd/$lib $K --shape enc720 --dist M --B 2 --op bwd --variants 0,100 --inner 4 --reps 5
done
