#!/bin/bash
# Round 6: tile-fed grad_value kernel, batch element major (tools/ab/gvt_bmajor: -DVNX_GVT_BATCH_MAJOR=1 on msda_d32_gvtiles.hip) against
# minor (base): encoder shapes, kbench cold + FETCH_SIZE
cd $GRAFT_REPO_ROOT
K=$GRAFT_REPO_ROOT/tools/kbench.bin
O=$GRAFT_REPO_ROOT/gpurun_out/r6_ab_enc.log
P=$GRAFT_REPO_ROOT/gpurun_out/r6_ab_enc_pmc
mkdir -p $P
: > $O
run() { echo "=== $1: ${@:2}" >> $O; if [ "$1" = base ]; then ${@:2} >> $O 2>&1; else LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/ab/$1 ${@:2} >> $O 2>&1; fi; }
for v in base gvt_bmajor; do
  run $v $K --shape enc360 --dist M --op bwd --variants 0 --check --inner 8
  run $v $K --shape enc720 --dist M --op bwd --variants 0 --inner 4 --reps 5
  run $v $K --shape enc720 --dist M --B 2 --op bwd --variants 0 --inner 4 --reps 5
  run $v $K --shape enc360 --dist M --dtype bf16 --op bwd --variants 0 --inner 8
done
cd /tmp && export TMPDIR=/tmp
for v in base gvt_bmajor; do
  if [ "$v" = base ]; then L=""; else L=$GRAFT_REPO_ROOT/tools/ab/$v; fi
  LD_LIBRARY_PATH=$L rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P -o ${v}_enc720 -- $K --shape enc720 --dist M --op bwd --variants 0 --cold-only --inner 2 --reps 2 > /dev/null 2> $P/${v}_enc720.err
  LD_LIBRARY_PATH=$L rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P -o ${v}_enc360 -- $K --shape enc360 --dist M --op bwd --variants 0 --cold-only --inner 2 --reps 2 > /dev/null 2> $P/${v}_enc360.err
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/r6_ab_enc_pmc/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            acc[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
    print(os.path.basename(f))
    for k, v in acc.items():
        if 'msda' in k or 'gv' in k:
            print('   ', k, 'n=%d' % len(v), 'FETCH_SIZE avg %.1f (KB units -> MB %.1f)' % (sum(v) / len(v), sum(v) / len(v) / 1024))
PY
grep -E "^===|variant" $O
rm -f $P/*.db
