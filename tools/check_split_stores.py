#!/usr/bin/env python3
"""Static check of the built device code (no GPU): kernels in which 16-byte stores came out as four dword stores.

Round 6 found the per-call 9 x 9 kernels of 'default' / 'raw' / 'partial_3' WITHOUT the reset observation issuing 4.0 store
instructions per env instead of 1.1 (25.7 us per call of 65 536 envs instead of 15): `*(float4 *)(char_ptr + 16u * j) = v` had
been split by the compiler in those instantiations, `float4_ptr[j] = v` is not.  The signature in the disassembly: runs of
`global_store_dword v[a:b], vN, off offset:12`.  Kernels that store scalars on purpose (the `<false>` = unaligned forms of the
clock-grid kernels, 'positions') are listed too; anything else is a regression.

    python tools/check_split_stores.py            (after `make -C wurm_amd/csrc`)"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
EXPECTED = ('grid_step_kernel<false>', 'grid_rollout_kernel<false>', 'grid_flush_kernel<false>', 'gridworld_lane_step_kernel<3,')
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for obj in sorted(glob.glob(os.path.join(ROOT, 'wurm_amd', 'csrc', '_build', '*.o'))):
        name = os.path.basename(obj)
        shutil.copy(obj, os.path.join(tmp, name))
        subprocess.run([OBJDUMP, '--offloading', name], cwd=tmp, capture_output=True)
        co = glob.glob(os.path.join(tmp, name + '.*gfx950'))
        if not co:
            continue
        dis = subprocess.run([OBJDUMP, '-d', co[0]], capture_output=True, text=True).stdout
        cur, cnt = None, {}
        for ln in dis.split('\n'):
            m = re.match(r'^[0-9a-f]+ <(_Z[^>]+)>:', ln)
            if m:
                cur = m.group(1)
            elif cur and re.search(r'global_store_dword v\[\d+:\d+\], v\d+, off offset:12\b', ln):
                cnt[cur] = cnt.get(cur, 0) + 1
        for k, n in sorted(cnt.items()):
            if n < 2:
                continue
            dem = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
            ok = any(e in dem for e in EXPECTED)
            bad += not ok
            print(f"{'expected  ' if ok else 'SPLIT     '}{name:20s} {n:3d}  {dem[:120]}")
print('kernels with unexpectedly split 16-byte stores:', bad)
sys.exit(1 if bad else 0)
