import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from wurm_amd.envs import MultiSnake
dev = torch.device('cuda:0')
N, K, S = 4096, 4, 25
env = MultiSnake(N, K, S, device=dev, seed=0)
T = 300
a = torch.randint(8, (T, K, N), device=dev)
keys = [f'agent_{i}' for i in range(K)]
acts = [dict(zip(keys, a[t].unbind(0))) for t in range(T)]
for t in range(100):
    o, r, d, info = env.step(acts[t]); env.reset(d['__all__'], return_observations=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(100, 200):
    o, r, d, info = env.step(acts[t]); env.reset(d['__all__'], return_observations=False)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print('host issue us/iter', th / 100 * 1e6, 'total', (time.perf_counter() - t0) / 100 * 1e6)
pr = cProfile.Profile()
pr.enable()
for t in range(200, 300):
    o, r, d, info = env.step(acts[t]); env.reset(d['__all__'], return_observations=False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
