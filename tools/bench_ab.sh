#!/bin/bash
# A/B of PRODUCT-library builds on the headline metric itself (the bench's fwd + bwd step), ~15 s of GPU per build:
#   python -c "from vnext_amd import build as b; b.build_hip(out='tools/ab/<name>/libvnext_hip.so', defines=('VNX_...=...',))"
#   gpurun -- 'bash tools/bench_ab.sh base <name> ... base'
# `base` = the in-tree library; every other name is tools/ab/<name>/libvnext_hip.so, loaded through VNX_HIP_LIB.
# (Round 6: backward-only kbench runs and the step disagree about the unit split of the paired backward -- DESIGN.md 3.3g;
#  the kernel spans the line prints come from the development library's stamps and do not follow the build under test.)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then L=""; else L=$GRAFT_REPO_ROOT/tools/ab/$v/libvnext_hip.so; fi
  VNX_HIP_LIB=$L python bench.py --no-cpu --no-model --no-cases --no-warm > gpurun_out/bench_ab_$v.json 2> gpurun_out/bench_ab_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_ab_$v.json").read().strip().splitlines()[-1])
print("$v", "Gpoints/s %.3f  us/step %.2f  fwd alone %.2f  bwd alone %.2f" % (d["value"], d["ms_per_step"]*1e3, d["roofline"]["us_per_launch"], d["roofline_bwd"]["us_per_launch"]))
PY
done
