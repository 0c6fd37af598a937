#!/usr/bin/env python3
"""Parses rocprofv3 --pmc counter_collection CSVs: per kernel name and grid size, the mean counter value per launch.
usage: parse_pmc.py <csv> [<csv> ...]  -> JSON on stdout"""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row['Kernel_Name']
            if 'wurm::' not in name and 'copy' not in name.lower() and 'elementwise' not in name:
                continue
            key = f"{name[:90]}|grid={row.get('Grid_Size', row.get('Grid_Size_X', '?'))}"
            acc[key][row['Counter_Name']].append((int(row.get('Dispatch_Id', 0)), float(row['Counter_Value'])))
out = {}
for key, counters in acc.items():
    out[key] = {}
    for c, dv in counters.items():
        v = [x for _, x in sorted(dv)]  # in dispatch order: tools/make_traffic_json.py splits runs of one kernel by position
        out[key][c] = {'launches': len(v), 'mean': sum(v) / len(v), 'min': min(v), 'max': max(v), 'values': v}
print(json.dumps(out, indent=1))
