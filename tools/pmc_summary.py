import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if not any(x in k for x in ('c64', 'wgrad_kernel', 'conv3x3_kernel', 'bn_bwd')):
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
