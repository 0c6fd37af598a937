#!/usr/bin/env python3
"""rocprofv3 kernel-trace CSV -> mean duration per (kernel, grid size): separates the configs that share a kernel.
usage: trace_summary.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if 'wurm::' not in r['Kernel_Name']:
            continue
        key = (r['Kernel_Name'].replace('void ', '')[:60], int(r['Grid_Size_X']), int(r['Workgroup_Size_X']),
               r['LDS_Block_Size'], r['VGPR_Count'], r.get('Scratch_Size', ''))
        acc[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print(f"{'kernel':60s} {'grid':>9s} {'wg':>4s} {'lds':>6s} {'vgpr':>5s} {'scr':>4s} {'calls':>6s} {'mean_us':>9s} {'min_us':>8s} {'max_us':>8s}")
for k, v in sorted(acc.items()):
    v2 = sorted(v)[len(v) // 10: len(v) - len(v) // 10] or v  # trim warm-up outliers
    print(f'{k[0]:60s} {k[1]:9d} {k[2]:4d} {k[3]:>6s} {k[4]:>5s} {k[5]:>4s} {len(v):6d} {sum(v2) / len(v2) / 1e3:9.2f} {min(v) / 1e3:8.2f} {max(v) / 1e3:8.2f}')
