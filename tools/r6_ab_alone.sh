#!/bin/bash
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
O=gpurun_out/r6_ab_alone.log
: > $O
run() { echo "=== $1: ${@:2}" >> $O; if [ "$1" = base ]; then ${@:2} >> $O 2>&1; else LD_LIBRARY_PATH=tools/ab/$1 ${@:2} >> $O 2>&1; fi; }
for v in base alone2cu alone2cu640; do
  for B in 1 2 5 10; do
    run $v $K --shape dec360 --dist U --B $B --op fbwd --variants 0
  done
  run $v $K --shape dec360 --dist M --B 10 --op fbwd --variants 0
done
grep -E "^===|variant" $O
