#!/bin/bash
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
O=gpurun_out/r6_ab_pair4.log
: > $O
run() { echo "=== $1: ${@:2}" >> $O; if [ "$1" = base ]; then ${@:2} >> $O 2>&1; else LD_LIBRARY_PATH=tools/ab/$1 ${@:2} >> $O 2>&1; fi; }
for v in base r5form; do
  run $v $K --shape dec360 --dist U --op both --variants 0 --check
  run $v $K --shape dec360 --dist M --op both --variants 0 --check
  run $v $K --shape dec360 --dist U --B 10 --op both --variants 0
  run $v $K --shape dec360 --dist U --B 2 --op both --variants 0
  run $v $K --shape dec720 --dist U --op both --variants 0 --inner 8
  run $v $K --shape dec720 --dist U --B 2 --op both --variants 0 --inner 8
  run $v $K --shape dec360 --dist U --dtype bf16 --op bwd --variants 0
  run $v $K --shape dec360 --dist U --op fbwd --variants 0
  run $v $K --shape dec360 --dist U --B 10 --op fbwd --variants 0
done
timeout 1200 python -m pytest tests/test_msda_gvdirect.py tests/test_msda_gpu.py tests/test_fused.py -x -q -m gpu > gpurun_out/r6_ab_pair_pytest.log 2>&1
tail -3 gpurun_out/r6_ab_pair_pytest.log
grep -E "^===|variant|step" $O
