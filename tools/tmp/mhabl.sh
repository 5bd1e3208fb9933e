cd $GRAFT_REPO_ROOT
for ab in mh1 mh2 mh3 mh7 mh4; do
  echo "== $ab"
  VNX_HIP_DEV_LIB=$GRAFT_REPO_ROOT/tools/ab/$ab/libvnext_hip_dev.so timeout 120 python tools/time_mask_head.py 706 703 710 2>&1 | grep "mask head" | grep -v train | cut -c1-70
done
