"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-kernel HBM-side bytes per launch.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/rNN_traffic.json

Units and corrections (MI355X_MICROARCH.md, "HBM"): both derived counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide (16 B/lane) reads at 64 B, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
Infinity-Cache hits are included (the counters sit on the L2's fabric side)."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0].strip()


def collect(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            a = acc[short(r['Kernel_Name'])]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    return acc


def main():
    fetch = collect(sys.argv[1], 'FETCH_SIZE')
    write = collect(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        fb = 2.0 * 1024.0 * f / nf if nf else None
        wb = 1024.0 * w / nw if nw else None
        out[k] = {'launches': nf or nw, 'fetch_bytes_per_launch': fb and round(fb), 'write_bytes_per_launch': wb and round(wb),
                  'hbm_bytes_per_launch': round((fb or 0) + (wb or 0))}
    json.dump({'unit': 'bytes per launch (FETCH_SIZE x2 x1024 + WRITE_SIZE x1024)', 'kernels': out}, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
