import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import fuzz_parity, faulthandler
kind, seed, cases = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.RandomState(seed)
fn = getattr(fuzz_parity, 'fuzz_' + kind)
for i in range(cases):
    st = rng.get_state()
    try:
        d = fn(rng)
    except AssertionError as e:
        print('MISMATCH', str(e)[:300]); break
    print(i, d, flush=True)
print('done', kind)
