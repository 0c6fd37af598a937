"""Which kernels run in the env loops besides the library's own?  A rocprofv3 --kernel-trace --stats target: every loop runs
exactly 40 iterations after its set-up, so a torch kernel with a call count that is a multiple of 40 is a host-side op
hiding between two launches (round 5: MultiSnake.rollout launched a copy and a fill per call, 10 us between launches)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from wurm_amd.envs import SingleSnake, SimpleGridworld, MultiSnake
dev = torch.device('cuda:0')
IT = 40


def single(N, S, mode, cls=SingleSnake, **kw):
    env = cls(N, S, observation_mode=mode, device=dev, seed=0, **kw)
    a = torch.randint(4, (IT, N), device=dev)
    tape = torch.randint(4, (IT, 4, N), device=dev)
    torch.cuda.synchronize()
    for t in range(IT):
        o = env.step(a[t]); env.reset(o[2])
    for t in range(IT):
        env.rollout(tape[t])
    torch.cuda.synchronize()


def multi(N, K, S, **kw):
    env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
    a = torch.randint(8, (IT, K, N), device=dev)
    dicts = [{f'agent_{i}': a[t, i] for i in range(K)} for t in range(IT)]
    tape = torch.randint(8, (IT, 4, K, N), device=dev)
    torch.cuda.synchronize()
    for t in range(IT):
        o = env.step(dicts[t]); env.reset(o[2]['__all__'])
    for t in range(IT):
        o = env.step(dicts[t]); env.reset(o[2]['__all__'], return_observations=False)
    for t in range(IT):
        env.rollout(tape[t])
    torch.cuda.synchronize()


single(512, 9, 'partial_2')
single(65536, 9, 'partial_2')
single(8192, 36, 'default')
single(64, 9, 'default', cls=SimpleGridworld, start_location=(4, 4))
single(65536, 9, 'default', cls=SimpleGridworld, start_location=(4, 4))
multi(4096, 4, 25)
multi(4096, 4, 25, respawn_mode='any', food_mode='random_rate', boost_cost_prob=0.25, observation_mode='partial_5', food_on_death_prob=0.33, food_rate=2.5e-4)
multi(512, 2, 12)
multi(1024, 10, 36, boost=True, respawn_mode='any')
print('done')
