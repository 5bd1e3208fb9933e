#!/bin/bash
cd $GRAFT_REPO_ROOT
K=./tools/kbench.bin
O=gpurun_out/r6_ab_fslab.log
: > $O
run() { echo "=== ${@}" >> $O; ${@} >> $O 2>&1; }
run $K --shape enc360 --dist M --op fbwd --variants 0,733 --check --inner 8
run $K --shape enc720 --dist M --op fbwd --variants 0,733 --inner 4 --reps 5
run $K --shape enc720 --dist M --B 2 --op fbwd --variants 0,733 --inner 4 --reps 5
run $K --shape enc360 --dist M --B 10 --op fbwd --variants 0,733 --inner 4 --reps 5
timeout 1500 python -m pytest tests/test_msda_fused.py tests/test_msda_slab.py tests/test_msda_gvtiles.py tests/test_transformer.py -x -q -m gpu > gpurun_out/r6_ab_fslab_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r6_ab_fslab_pytest.log
grep -E "^===|variant" $O
