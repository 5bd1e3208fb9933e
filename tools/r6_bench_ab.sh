#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in base p_f44 p_f21 p_f81 base; do
  if [ "$v" = base ]; then L=""; else L=$GRAFT_REPO_ROOT/tools/ab/$v/libvnext_hip.so; fi
  VNX_HIP_LIB=$L python bench.py --no-cpu --no-model --no-cases --no-warm > gpurun_out/r6_bench_ab_$v.json 2> gpurun_out/r6_bench_ab_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6_bench_ab_$v.json").read().strip().splitlines()[-1])
print("$v", "Gpoints/s %.3f  us/step %.2f  fwd %.2f (span %.2f)  bwd %.2f (span %.2f)" % (d["value"], d["ms_per_step"]*1e3, d["roofline"]["us_per_launch"], d["roofline"].get("us_kernel_span") or 0, d["roofline_bwd"]["us_per_launch"], d["roofline_bwd"].get("us_kernel_span") or 0))
PY
done
