import csv, glob, sys, collections
tag = sys.argv[1]
for d in sorted(glob.glob(f'gpurun_out/pmc_{tag}_p*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if 'igemm' not in r['Kernel_Name']: continue
            a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
        for k, (n, v) in sorted(agg.items()):
            print(f"{k:32s} n={n:3d} avg={v/n:16.1f}")
