#!/bin/bash
# round 6: the paired backward with three workgroups per CU (channel parts of the grad_value unit) against round 5's form
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
O=gpurun_out/r6_ab_pair.log
: > $O
run() { echo "=== $1: ${@:2}" >> $O; if [ "$1" = base ]; then ${@:2} >> $O 2>&1; else LD_LIBRARY_PATH=tools/ab/$1 ${@:2} >> $O 2>&1; fi; }
for v in base r5form rows768; do
  run $v $K --shape dec360 --dist U --op both --variants 0 --check
  run $v $K --shape dec360 --dist M --op bwd --variants 0 --check
  run $v $K --shape dec360 --dist U --B 10 --op bwd --variants 0
  run $v $K --shape dec720 --dist U --op bwd --variants 0 --inner 8
done
timeout 900 python -m pytest tests/test_msda_gvdirect.py tests/test_msda_gpu.py -x -q -m gpu > gpurun_out/r6_ab_pair_pytest.log 2>&1
tail -3 gpurun_out/r6_ab_pair_pytest.log
grep -E "^===|bwd|both|step" $O | head -150
