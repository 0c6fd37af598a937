#!/usr/bin/env python3
"""prints `name  wall us  gpu us` per line of a tools/bench_configs.py output file"""
import json
import sys
for ln in open(sys.argv[1]):
    try:
        d = json.loads(ln)
    except ValueError:
        continue
    k = list(d)[0]
    v = d[k]
    print(f'{k:70s} {v["us_per_batch_step"]:8.2f} {v["gpu_us_per_batch_step"]:8.2f}')
