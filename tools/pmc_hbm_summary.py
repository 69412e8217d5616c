"""Per-kernel HBM traffic from separate rocprofv3 ``--pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` passes (KiB per dispatch, raw:
FETCH_SIZE needs the x2 gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md "HBM").

    python tools/pmc_hbm_summary.py DIR_FETCH DIR_WRITE > profiles/rNN_pmc_hbm_traffic.json"""
import csv
import glob
import json
import sys
from collections import defaultdict

out = defaultdict(dict)
for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            short = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
            a = acc[short][r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    for k, cs in acc.items():
        for c, (tot, n) in cs.items():
            out[k][c] = {'avg_KiB': tot / n, 'dispatches': n}
json.dump(out, sys.stdout, indent=1)
