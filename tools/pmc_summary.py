"""Sum rocprofv3 --pmc counter_collection CSVs per kernel (development tool)."""
import collections, csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_tile'
key = sys.argv[2] if len(sys.argv) > 2 else ''
for f in sorted(glob.glob(pat + '/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    for k, v in agg.items():
        if key in k:
            print(f.split('/')[-1], k, 'dispatches', len(disp[k]))
            for c, x in sorted(v.items()):
                print(f'    {c:28s} {x:16.0f}  per dispatch {x/len(disp[k]):14.0f}')
