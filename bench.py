#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched environment step on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): SingleSnake, 512 envs per GPU, 9x9, observation_mode='partial_2', uniform random
actions, the reference's test loop `obs, r, done, info = env.step(a[t]); env.reset(done)`
(tests/test_single_snake_env.py:24-31 in oscarknagg/wurm).

One bench STEP = one batch-step: every env of the batch advanced by one step(), observed, and reset if done.
K steps are executed through the fused rollout entry point (wurm_single_rollout, `--chunk` batch-steps per
launch, bit-identical to K step()/reset() call pairs — tests/test_hip_vs_oracle.py); inputs (state, action
tape) are resident in HBM before the timed region; observations, rewards and dones of every step are written
to HBM inside it.  `value` = env-steps of all ranks / max-over-ranks wall time.

Multi-GPU (weak scaling): one process per GPU, each stepping its own contiguous block of env ids
(env_offset = rank * num_envs) with no data-path collective — envs never interact; RCCL is used only for the
barrier, the max-over-ranks time and the summed episode statistics.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

SIZE = 9
OBS_MODE = 'partial_2'
OBS_ELEMS = 75
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def algorithmic_bytes_per_env_step(size: int, obs_elems: int) -> int:
    """read 3 channels + write head,body (fp32) + observation + ~39 B of per-env scalars."""
    return 20 * size * size + 4 * obs_elems + 39


def cpu_baseline(num_envs: int, budget_s: float = 12.0):
    """The CPU oracle (scalar C port of the reference's algorithm, 1 thread) on the same workload, bounded."""
    import numpy as np
    from oracle import oracle
    envs = np.zeros((num_envs, 3, SIZE, SIZE), np.float32)
    oracle.single_reset(envs, np.ones(num_envs, np.uint8), 'none', seed=0, call=0)
    rng = np.random.RandomState(0)
    chunk, steps, call = 100, 0, 1
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        actions = rng.randint(0, 4, size=(chunk, num_envs)).astype(np.int64)
        oracle.single_rollout(envs, actions, OBS_MODE, seed=0, call0=call)
        call += 2 * chunk
        steps += chunk
    dt = time.perf_counter() - t0
    return {'value': num_envs * steps / dt, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'oracle/single_snake.c (scalar C restatement, 1 thread): SingleSnake {num_envs}x{SIZE}x{SIZE} '
                      f'{OBS_MODE}, {steps} batch-steps of step+observe+reset in {dt:.1f} s; the reference torch-CPU '
                      f'path itself measured 52 372 env-steps/s on 8 vCPU in the build container (BASELINE.md §2)'}


def run_rollouts(env, actions, first, steps, chunk, events=None):
    """`steps` batch-steps starting at row `first` of the action tape, `chunk` per launch."""
    done_eps = None
    for c in range(0, steps, chunk):
        n = min(chunk, steps - c)
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        out = env.rollout(actions[first + c:first + c + n])
        if events is not None:
            e1.record()
            events.append((e0, e1, n))
        done_eps = out
    return done_eps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=262144, help='timed batch-steps (K)')
    ap.add_argument('--warmup', type=int, default=4096, help='untimed batch-steps (W)')
    ap.add_argument('--num-envs', type=int, default=512, help='envs per GPU (BASELINE configs[1]: 512)')
    ap.add_argument('--chunk', type=int, default=1024, help='batch-steps per rollout launch')
    ap.add_argument('--no-extra', action='store_true', help='skip the secondary measurements')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print(f'# note: --gpus {args.gpus} but WORLD_SIZE={world}; using {n_gpus}', file=sys.stderr)
    device = torch.device('cuda', local_rank)

    from wurm_amd.envs import SingleSnake
    N, K, W = args.num_envs, args.steps, args.warmup
    env = SingleSnake(num_envs=N, size=SIZE, observation_mode=OBS_MODE, device=device, seed=0, env_offset=rank * N)
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    actions = torch.randint(4, (W + K, N), generator=gen, device=device, dtype=torch.int64)

    def barrier():
        if distributed:
            dist.barrier()

    run_rollouts(env, actions, 0, W, args.chunk)
    # one more untimed launch with the shape of the timed launches (a small --steps / --warmup would otherwise time the
    # allocator's first encounter with these output sizes); it is not counted in W
    prime = torch.randint(4, (min(args.chunk, max(K, 1)), N), generator=gen, device=device, dtype=torch.int64)
    env.rollout(prime)
    del prime
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    events = []
    t0 = time.perf_counter()
    run_rollouts(env, actions, W, K, args.chunk, events)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    total_env_steps = N * K * n_gpus

    # dominant kernel: rollout_s9_kernel<4>; per-launch duration from HIP events on the launch stream
    full = [(e0.elapsed_time(e1) * 1e-3, n) for e0, e1, n in events if n == args.chunk] or \
           [(e0.elapsed_time(e1) * 1e-3, n) for e0, e1, n in events]
    avg_launch_s = sum(d for d, _ in full) / len(full)
    bytes_per_launch = algorithmic_bytes_per_env_step(SIZE, OBS_ELEMS) * N * full[0][1]
    achieved = bytes_per_launch / avg_launch_s / 1e9

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f'rollout_{N}x{SIZE}_chunk{args.chunk}')
            except Exception:
                traffic = None
        line = {
            'metric': 'env_steps_per_s', 'value': total_env_steps / elapsed, 'unit': 'env-steps/s',
            'n_gpus': n_gpus, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'i32',
            'data': 'synthetic',
            'config': {'workload': f'SingleSnake num_envs={N}/GPU size={SIZE} obs={OBS_MODE} random actions, '
                                   f'step+observe+reset per batch-step, fused rollout launches of {args.chunk} '
                                   f'batch-steps', 'num_envs_per_gpu': N, 'global_num_envs': N * n_gpus,
                       'size': SIZE, 'observation_mode': OBS_MODE, 'parallelism': f'env-batch split x{n_gpus}',
                       'state_dtype': 'fp32 NCHW (exact integers)', 'chunk': args.chunk},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'wurm::rollout_s9_kernel<4> (9x9 SingleSnake, partial_n crop, RNG mode; one wave per env)', 'avg_launch_ms': avg_launch_s * 1e3,
                         'algorithmic_bytes_per_env_step': algorithmic_bytes_per_env_step(SIZE, OBS_ELEMS),
                         'env_steps_per_launch': N * full[0][1],
                         'note': 'this config is issue-bound, not HBM-bound: 512 envs = 512 lone waves on 1024 SIMDs, '
                                 '~60 instructions per env-step at 5-6 cycles each (DESIGN.md §4.4); `traffic` is '
                                 'below the algorithmic bytes because the env state never leaves the registers'},
        }
        if n_gpus == 1 and not args.no_extra:
            line['extra'] = extra_measurements(device)
        if n_gpus == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(N)
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


def extra_measurements(device):
    """Secondary numbers (not the headline): the per-call Python API on the same workload and one GPU's share of
    BASELINE configs[2] (8192 envs)."""
    from wurm_amd.envs import SingleSnake
    out = {}
    # (a) step()/reset() call pairs from Python — what a policy-in-the-loop caller pays
    N, T = 512, 2000
    env = SingleSnake(num_envs=N, size=SIZE, observation_mode=OBS_MODE, device=device, seed=0)
    actions = torch.randint(4, (T + 200, N), device=device, dtype=torch.int64)
    for t in range(200):
        _, _, d, _ = env.step(actions[t])
        env.reset(d, return_observations=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(200, 200 + T):
        _, _, d, _ = env.step(actions[t])
        env.reset(d, return_observations=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['per_call_api_512'] = {'value': N * T / dt, 'unit': 'env-steps/s', 'us_per_batch_step': dt / T * 1e6,
                               'what': 'Python loop of env.step(a); env.reset(done) (2 launches per batch-step)'}
    # (b) one GPU's share of configs[2]: 8192 envs
    N, T, chunk = 8192, 2048, 128
    env = SingleSnake(num_envs=N, size=SIZE, observation_mode=OBS_MODE, device=device, seed=0)
    actions = torch.randint(4, (chunk + T, N), device=device, dtype=torch.int64)
    env.rollout(actions[:chunk])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(chunk, chunk + T, chunk):
        env.rollout(actions[c:c + chunk])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gbs = algorithmic_bytes_per_env_step(SIZE, OBS_ELEMS) * N * T / dt / 1e9
    out['rollout_8192'] = {'value': N * T / dt, 'unit': 'env-steps/s', 'achieved_GBs': gbs,
                           'frac_of_hbm_peak': gbs / HBM_PEAK_GBS,
                           'what': 'BASELINE configs[2] per-GPU share (65536/8), fused rollout, chunk 128'}
    del env, actions
    # (c) BASELINE configs[4] shape: SingleSnake 8192 x 36 x 36, default RGB observation, fused rollout
    N, S, chunk, reps = 8192, 36, 16, 6
    env = SingleSnake(num_envs=N, size=S, observation_mode='default', device=device, seed=0)
    actions = torch.randint(4, (chunk * (reps + 1), N), device=device, dtype=torch.int64)
    env.rollout(actions[:chunk])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(1, reps + 1):
        env.rollout(actions[r * chunk:(r + 1) * chunk])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eps = N * chunk * reps / dt
    per = algorithmic_bytes_per_env_step(S, 3 * S * S)
    out['rollout_cfg5_8192x36_default'] = {
        'value': eps, 'unit': 'env-steps/s', 'achieved_GBs': per * eps / 1e9, 'obs_write_GBs': 12 * S * S * eps / 1e9,
        'frac_of_hbm_peak': per * eps / 1e9 / HBM_PEAK_GBS,
        'what': 'BASELINE configs[4]: SingleSnake 8192x36x36 default-RGB obs, fused rollout, chunk 16 '
                '(algorithmic 41 511 B per env-step; the fused path only writes the 15 552 B observation)'}
    del env, actions
    # (d) BASELINE configs[3]: MultiSnake 4096 x 25 x 25, 4 agents, constructor defaults ('full' obs), fused rollout
    from wurm_amd.envs import MultiSnake
    N, K, S, chunk, reps = 4096, 4, 25, 16, 6
    env = MultiSnake(N, K, S, device=device, seed=0)
    actions = torch.randint(8, (chunk * (reps + 1), K, N), device=device, dtype=torch.int64)
    env.rollout(actions[:chunk])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(1, reps + 1):
        env.rollout(actions[r * chunk:(r + 1) * chunk])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eps = N * chunk * reps / dt
    per = 8 * (1 + 2 * K) * S * S + 12 * K * S * S + 40 * K
    out['multi_rollout_cfg4_4096x25_k4_full'] = {
        'value': eps, 'unit': 'env-steps/s', 'achieved_GBs': per * eps / 1e9, 'obs_write_GBs': 12 * K * S * S * eps / 1e9,
        'frac_of_hbm_peak': per * eps / 1e9 / HBM_PEAK_GBS,
        'what': 'BASELINE configs[3]: MultiSnake 4096x25x25, 4 agents, defaults, step+observe+reset(__all__) per '
                'batch-step, fused rollout chunk 16 (algorithmic 75 160 B per env-step; reference torch-CPU: 3 280 '
                'env-steps/s)'}
    del env, actions
    # (e) the acting loop with the policy inside the env kernel: MLP 75->64->64->{4,1} + sampling
    from wurm_amd.agents import FeedforwardAgent, pack_policy_params
    N, T, reps = 512, 256, 8
    torch.manual_seed(0)
    env = SingleSnake(num_envs=N, size=SIZE, observation_mode=OBS_MODE, device=device, seed=0)
    params = pack_policy_params(FeedforwardAgent(4, 2, 64, OBS_ELEMS).to(device))
    state = env.reset()
    state = env.policy_rollout(params, state, T, check=False)['state']
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        state = env.policy_rollout(params, state, T, check=False)['state']
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['policy_rollout_512'] = {
        'value': N * T * reps / dt, 'unit': 'env-steps/s',
        'what': 'policy forward (random-init FeedforwardAgent) + Categorical sample + step + observe + reset per '
                'env-step, fused in one kernel, launches of 256 batch-steps'}
    return out


if __name__ == '__main__':
    main()
