#!/usr/bin/env python3
"""Per-kernel resource table from the compiler's own remarks (-Rpass-analysis=kernel-resource-usage): VGPRs, AGPRs,
SGPRs, spills, scratch, LDS, occupancy (waves per SIMD).  No GPU needed.  usage: tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'wurm_amd', 'csrc')
FLAGS = '-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage'.split()
rows = []
for src in sorted(f for f in os.listdir(CSRC) if f.endswith('.hip')):
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', *FLAGS, '-c', src, '-o', '/dev/null'],
                       cwd=CSRC, capture_output=True, text=True)
    cur = None
    for ln in r.stderr.splitlines():
        m = re.search(r'remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass', ln) or re.search(r'remark:\s+(.*?) \[-Rpass', ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith('Function Name:') or t.startswith('Name:'):
            name = t.split(':', 1)[1].strip()
            dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            cur = {'kernel': dem.replace('wurm::', '').replace('(wurm::StepArgs)', '').replace('(wurm::MultiArgs)', ''), 'file': src}
            rows.append(cur)
        elif cur is not None and ':' in t:
            k, v = t.split(':', 1)
            cur[k.strip()] = v.strip()
cols = [('VGPRs', 'vgpr'), ('AGPRs', 'agpr'), ('TotalSGPRs', 'sgpr'), ('SGPRs Spill', 'sgpr_spill'), ('VGPRs Spill', 'vgpr_spill'),
        ('ScratchSize [bytes/lane]', 'scratch'), ('LDS Size [bytes/block]', 'lds_static'), ('Occupancy [waves/SIMD]', 'waves/simd')]
print(f"{'kernel':78s} " + ' '.join(f'{c[1]:>10s}' for c in cols))
for r in rows:
    print(f"{r['kernel'][:78]:78s} " + ' '.join(f"{r.get(c[0], '-'):>10s}" for c in cols))
