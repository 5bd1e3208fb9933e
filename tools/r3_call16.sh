cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
$K --shape dec360 --dist U --op bwd --variants 0 --timeline > gpurun_out/c16_plain.log 2>&1
cp vnext_amd/lib/libvnext_hip_stamps.so vnext_amd/lib/libvnext_hip.so
$K --shape dec360 --dist U --op bwd --variants 0 --timeline > gpurun_out/c16_stamps.log 2>&1
