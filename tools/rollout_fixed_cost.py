"""Fixed per-launch cost of the fused rollout: time per launch against the tape length (SingleSnake N x 9 x 9 partial_2)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wurm_amd.envs import SingleSnake
dev = torch.device('cuda:0')
def time_rollout(N, T, reps=8):
    env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device=dev, seed=0)
    actions = torch.randint(4, (T, N), device=dev, dtype=torch.int64)
    for _ in range(2):
        env.rollout(actions.clone())
    acts = [actions.clone() for _ in range(reps)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for a in acts:
        out = env.rollout(a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for N, Ts in [(65536, [16, 32, 64, 128]), (8192, [32, 64, 128, 256, 512])]:
    for T in Ts:
        ms = time_rollout(N, T)
        print(json.dumps({'N': N, 'T': T, 'ms': round(ms, 4), 'us_per_step': round(ms * 1e3 / T, 3)}), flush=True)
