"""Aggregate a rocprofv3 ``--pmc`` run (``*_counter_collection.csv``) per kernel: mean counter value per dispatch.

    python tools/pmc_summary.py DIR [DIR ...] > profiles/rNN_pmc_*.json
When SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE are both present, adds
  MfmaUtil_percent = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)   (rocprofv3's MfmaUtil formula; on gfx950
                     GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs - it is 7.5-8x the kernel duration in cycles)
  effective_clock_MHz = GRBM_GUI_ACTIVE / 8 / kernel duration: the power-limited shader clock the kernel actually ran at."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in sys.argv[1:]:
        for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
            per_dispatch = defaultdict(dict)
            for r in csv.DictReader(open(f)):
                per_dispatch[(r['Dispatch_Id'], r['Kernel_Name'])][r['Counter_Name']] = float(r['Counter_Value'])
            dur = {}
            for r in csv.DictReader(open(f)):
                dur[(r['Dispatch_Id'], r['Kernel_Name'])] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
            for key, cs in per_dispatch.items():
                cs['_duration_ns'] = dur[key]
            for (_, name), cs in per_dispatch.items():
                short = name.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
                for c, v in cs.items():
                    a = acc[short][c]
                    a[0] += v
                    a[1] += 1
    out = {}
    for k, cs in acc.items():
        o = {c: {'mean_per_dispatch': v[0] / v[1], 'dispatches': v[1]} for c, v in cs.items()}
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs and cs['GRBM_GUI_ACTIVE'][0] > 0:
            o['MfmaUtil_percent'] = 100.0 * cs['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (cs['GRBM_GUI_ACTIVE'][0] / 8 * 1024)
            o['effective_clock_MHz'] = cs['GRBM_GUI_ACTIVE'][0] / 8 / cs['_duration_ns'][0] * 1e3
        out[k] = o
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
