#!/usr/bin/env python3
"""Instruction mix of one loop of one kernel in a hipcc -S listing.
usage: tools/isa_loop_mix.py file.s <kernel-substring> [loop-header-label]  (without a label: the list of loops)"""
import collections
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else None
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and kern in l and l.rstrip().endswith(tuple([':']) ) or (l.startswith('_Z') and kern in l and ': ' in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks = []  # (label, headers[list], instrs)
cur = None
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?$', l)
    if m:
        cur = {'label': m.group(1), 'hdr': [], 'ins': []}
        blocks.append(cur)
        c = m.group(2) or ''
        h = re.search(r'Header=(BB\d+_\d+)', c)
        if h: cur['hdr'].append('.L' + h.group(1))
        if 'Loop Header' in c: cur['hdr'].append(m.group(1))
        continue
    if cur is None:
        continue
    c = l.strip()
    if c.startswith(';'):
        h = re.search(r'Parent Loop (BB\d+_\d+)', c)
        if h: cur['hdr'].append('.L' + h.group(1))
        if 'Loop Header' in c: cur['hdr'].append(cur['label'])
        continue
    if re.match(r'^\s+[a-z]', l):
        cur['ins'].append(c.split()[0])
loops = collections.OrderedDict()
for b in blocks:
    for h in set(b['hdr']):
        loops.setdefault(h, []).append(b)
if want is None:
    for h, bs in loops.items():
        print(h, 'blocks', len(bs), 'instructions', sum(len(b['ins']) for b in bs))
    sys.exit(0)
mix = collections.Counter()
for b in loops['.L' + want.lstrip('.L')] if not want.startswith('.L') else loops[want]:
    mix.update(b['ins'])
cls = collections.Counter()
for k, v in mix.items():
    c = 'salu' if k.startswith('s_') else 'lds' if k.startswith('ds_') else 'vmem' if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'valu'
    cls[c] += v
print(dict(cls), 'total', sum(mix.values()))
for k, v in mix.most_common(60):
    print(f'{v:5d} {k}')
