"""Static check of the BUILT device code (no GPU): no kernel may have its 16-byte observation stores split into four dword
stores by the compiler — round 6 found the per-call 9 x 9 kernels of 'default' / 'raw' / 'partial_3' without the reset observation
and every 64-envs-per-wave rollout instantiation doing that (4.0 store instructions per env instead of 1.1; tools/
check_split_stores.py explains the signature).  Needs the object files of `make -C wurm_amd/csrc` (what
`__graft_entry__.build()` leaves behind) and llvm-objdump; skipped where either is missing."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_has_its_16_byte_stores_split():
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('no llvm-objdump')
    if not glob.glob(os.path.join(ROOT, 'wurm_amd', 'csrc', '_build', '*.o')):
        pytest.skip('no object files (the library was not built in this tree)')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_split_stores.py')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert 'kernels with unexpectedly split 16-byte stores: 0' in r.stdout
