import csv, sys, glob, collections
rows = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        key = (r['Dispatch_Id'], r['Kernel_Name'].split('(')[0][-60:], r.get('Grid_Size', ''))
        rows.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
names = sorted({c for v in rows.values() for c in v})
print('kernel', *names, sep='\t')
for (d, k, gs), v in rows.items():
    print(f'{d}:{k}:{gs}', *[f'{v.get(c, 0):.4g}' for c in names], sep='\t')
