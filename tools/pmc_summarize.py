import csv, sys, glob, collections, re
rows = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).replace('void ', '').split('(')[0]
        key = (r['Dispatch_Id'], name[-60:], r.get('Grid_Size', ''))
        rows.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
names = sorted({c for v in rows.values() for c in v})
print('kernel', *names, sep='\t')
for (d, k, gs), v in rows.items():
    print(f'{d}:{k}:{gs}', *[f'{v.get(c, 0):.4g}' for c in names], sep='\t')
