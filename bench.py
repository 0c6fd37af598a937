#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched environment step on MI355X (BASELINE.json metric).

Default workload (BASELINE.json configs[1], `--workload cfg2`): SingleSnake, 512 envs per GPU, 9x9,
observation_mode='partial_2', uniform random actions, the reference's test loop
`obs, r, done, info = env.step(a[t]); env.reset(done)` (tests/test_single_snake_env.py:24-31,
experiments/speeds.py:30-44 in oscarknagg/wurm).

One bench STEP = ONE PASS OF THE HOT PATH OVER ONE BATCH OF SYNTHETIC INPUT = one launch of the fused rollout entry
point (wurm_single_rollout) over one (chunk, num_envs) block of the action tape, i.e. `chunk` iterations of the loop
above for every env of the batch (bit-identical to `chunk` step()/reset() call pairs: tests/test_hip_vs_oracle.py).
`--steps K --warmup W` therefore times exactly K launches of the stated shape after W untimed ones;
`ms_per_step` is per launch; env-steps = K * chunk * num_envs * n_gpus.  Inputs (state, action tape) are resident in
HBM before the timed region; observations, rewards, dones of every env-step are written to HBM inside it.
`value` = env-steps of all ranks / max-over-ranks wall time.

Multi-GPU: one process per GPU, each stepping its own contiguous block of global env ids (env_offset) with no
data-path collective — envs never interact; RCCL is used only for the barrier, the max-over-ranks time and the summed
episode statistics.  `python bench.py --gpus N` with N > 1 STARTS ITS OWN N RANKS (a fresh
`python -m torch.distributed.run` child, launched before this process touches the GPU) and relays rank 0's JSON
line; when the driver has already started the ranks (WORLD_SIZE / RANK in the environment) this process is one of them.
  --workload cfg2 : 512 envs per GPU (weak scaling; the default and the headline)
  --workload cfg3 : BASELINE configs[2] — 65 536 envs in total, split 65 536/N per rank (strong scaling)
Whatever the workload, the line also carries `cfg3_strong_scaling`: BASELINE configs[2] (65 536 envs split over the N
ranks, the multi-GPU configuration north_star names) timed with the same barrier / max-over-ranks protocol right after
the headline region — so that a `--gpus N` sweep of the default command measures the batch split, not only the
communication-free weak scaling of 512 envs per GPU.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIZE = 9
OBS_MODE = 'partial_2'
OBS_ELEMS = 75
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # what a float4 copy reaches on this part (same guide)
LANE_ROLLOUT_MIN_ENVS = 6144  # from here on rollouts of 9 x 9 run one env per LANE (wurm_amd/csrc/lane_rollout.hpp)
MAX_TAPE_BLOCKS = 160  # distinct (chunk, N) action blocks kept in HBM; longer runs cycle through them


def algorithmic_bytes_per_env_step(size: int, obs_elems: int) -> int:
    """read 3 channels + write head,body (fp32) + observation + ~39 B of per-env scalars."""
    return 20 * size * size + 4 * obs_elems + 39


# ------------------------------------------------------------------------------------------------ launcher

def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=128, help='timed rollout launches (K)')
    ap.add_argument('--warmup', type=int, default=8, help='untimed rollout launches (W)')
    ap.add_argument('--workload', choices=('cfg2', 'cfg3'), default='cfg2')
    ap.add_argument('--num-envs', type=int, default=None, help='envs per GPU (cfg2: 512) / in total (cfg3: 65536)')
    ap.add_argument('--chunk', type=int, default=None, help='batch-steps per rollout launch (cfg2: 1024)')
    ap.add_argument('--no-extra', action='store_true', help='skip the secondary measurements')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / collective plumbing only: gloo on CPU, the rollout launch replaced by a sleep')
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def free_port() -> int:
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args) -> int:
    """Starts `--gpus` fresh rank processes (this process has not touched the GPU and never will), relays rank 0's
    JSON line.  A failed child means a non-zero exit code."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__), '--worker']
    cmd += [a for a in sys.argv[1:] if a != '--worker']
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    for ln in r.stdout.splitlines():
        if not ln.startswith('{'):
            print(ln, file=sys.stderr)
    if r.returncode != 0 or not lines:
        print(f'bench.py: rank processes failed (exit code {r.returncode})', file=sys.stderr)
        return r.returncode or 1
    print(lines[-1])
    return 0


# ------------------------------------------------------------------------------------------------ CPU baselines

def cpu_baseline(num_envs: int, budget_s: float = 5.0):
    """The reference's algorithm on the host cores of this box, on a bounded sample of the cfg2 workload
    (step + observe + reset per batch-step, random actions).  Three legs (SURVEY.md §8(d) i-iii):
      value           torch-op restatement of the convolution-and-mask algorithm (oracle/torch_port.py), all cores
      c_port_1thread  scalar C restatement (oracle/single_snake.c), one thread
      c_port_allcores the same, env batch split over one thread per core
    The reference itself cannot travel to the GPU box; its own number measured in the build container is quoted."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from oracle.torch_port import TorchSingleSnake
    cores = os.cpu_count() or 1
    out = {}

    # (ii) torch-op restatement.  The reference's own number was taken at 8 torch threads; tiny tensors
    # and hundreds of OpenMP threads do not mix (each op pays the fork/join of the whole team), so the leg is timed at 1,
    # 8 and all cores, each bounded per batch-step, and `value` is the best of them (with its thread count as `cores`).
    def torch_leg(threads, budget):
        torch.set_num_threads(threads)
        env = TorchSingleSnake(num_envs, SIZE, OBS_MODE, seed=0)
        g = torch.Generator().manual_seed(0)
        _, _, d, _ = env.step(torch.randint(4, (num_envs,), generator=g))
        env.reset(d)
        steps, t0 = 0, time.perf_counter()
        while True:
            _, _, d, _ = env.step(torch.randint(4, (num_envs,), generator=g))
            env.reset(d)
            steps += 1
            if time.perf_counter() - t0 >= budget:
                break
        dt = time.perf_counter() - t0
        return {'value': num_envs * steps / dt, 'cores': threads, 'sample': f'{steps} batch-steps in {dt:.1f} s'}
    legs = [torch_leg(t, budget_s / 2) for t in sorted({1, min(8, cores), cores})]
    best = max(legs, key=lambda l: l['value'])
    torch_rate, torch_cores = best['value'], best['cores']
    torch_sample = f"{best['sample']}, torch {torch.__version__}, {best['cores']} threads (best of " + \
                   ', '.join(f"{l['cores']} thr: {l['value']:.3g}" for l in legs) + ' env-steps/s)'
    torch.set_num_threads(min(8, cores))

    # (iii) scalar C restatement: 1 thread, then one thread per core (ctypes releases the GIL during the call)
    def c_port(n_threads):
        bounds = [num_envs * i // n_threads for i in range(n_threads + 1)]
        shards = []
        for lo, hi in zip(bounds, bounds[1:]):
            e = np.zeros((hi - lo, 3, SIZE, SIZE), np.float32)
            if hi > lo:
                oracle.single_reset(e, np.ones(hi - lo, np.uint8), 'none', seed=0, call=0, env_offset=lo)
            shards.append((lo, hi, e))
        rng = np.random.RandomState(0)
        chunk, steps, call = 200, 0, 1
        pool = ThreadPoolExecutor(n_threads) if n_threads > 1 else None

        def run(shard, actions, call0):
            lo, hi, e = shard
            if hi > lo:
                oracle.single_rollout(e, np.ascontiguousarray(actions[:, lo:hi]), OBS_MODE, seed=0, call0=call0,
                                      env_offset=lo)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            actions = rng.randint(0, 4, size=(chunk, num_envs)).astype(np.int64)
            if pool:
                list(pool.map(lambda s: run(s, actions, call), shards))
            else:
                run(shards[0], actions, call)
            call += 2 * chunk
            steps += chunk
        dt = time.perf_counter() - t0
        if pool:
            pool.shutdown()
        return num_envs * steps / dt, steps, dt

    r1, s1, d1 = c_port(1)
    n_thr = max(1, min(cores, num_envs))
    rn, sn, dn = c_port(n_thr)
    cpu_model = 'unknown'
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                cpu_model = ln.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    out.update({
        'value': torch_rate, 'unit': 'env-steps/s', 'cores': torch_cores, 'kind': 'port', 'host_cores': cores,
        'torch_port_by_threads': legs,
        'sample': f'oracle/torch_port.py (torch-op restatement of the reference\'s conv-and-mask algorithm, bit-equal '
                  f'to the oracle: tests/test_torch_port.py): SingleSnake {num_envs}x{SIZE}x{SIZE} {OBS_MODE}, '
                  f'step+observe+reset per batch-step, {torch_sample}',
        'cpu_model': cpu_model,
        'c_port_1thread': {'value': r1, 'cores': 1, 'sample': f'oracle/single_snake.c, {s1} batch-steps in {d1:.1f} s'},
        'c_port_allcores': {'value': rn, 'cores': n_thr,
                            'sample': f'oracle/single_snake.c, env batch split over {n_thr} threads, {sn} batch-steps '
                                      f'in {dn:.1f} s'},
        'reference_measured_elsewhere': {'value': 52372, 'cores': 8,
                                         'sample': 'the real reference (torch-CPU, 8 vCPU Xeon 2.1 GHz, build '
                                                   'container), BASELINE.md §2 — cannot travel to the GPU box'},
    })
    return out


# ------------------------------------------------------------------------------------------------ worker

def worker(args) -> int:
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    distributed = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ  # started by torchrun (even with one rank)
    dist = None
    if distributed:
        import torch.distributed as dist
        if args.dry_run:
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    n_gpus = world
    if args.gpus != n_gpus and rank == 0:
        print(f'# note: --gpus {args.gpus} but WORLD_SIZE={world}; running {n_gpus} rank(s)', file=sys.stderr)
    device = torch.device('cpu') if args.dry_run else torch.device('cuda', local_rank)

    from wurm_amd.sharding import shard_range
    K, W = max(args.steps, 1), max(args.warmup, 0)
    if args.workload == 'cfg2':
        N = args.num_envs or 512
        offset, global_envs, scaling = rank * N, N * n_gpus, 'weak'
        chunk = args.chunk or 1024
    else:
        global_envs = args.num_envs or 65536
        offset, N = shard_range(global_envs, rank, n_gpus)
        scaling = 'strong'
        chunk = args.chunk or (64 if N > 16384 else 128)

    def barrier():
        if distributed:
            dist.barrier()

    def sync():
        if not args.dry_run:
            torch.cuda.synchronize()

    def timed_region(N, offset, chunk, K, W):
        """W untimed + exactly K timed rollout launches of (chunk, N) bracketed by barrier + synchronize on both sides;
        returns (max-over-ranks seconds, env-steps of all ranks, HIP-event average launch seconds on this rank, blocks)"""
        blocks = min(W + K, MAX_TAPE_BLOCKS)
        if args.dry_run:
            env, tape = None, None
        else:
            from wurm_amd.envs import SingleSnake
            env = SingleSnake(num_envs=N, size=SIZE, observation_mode=OBS_MODE, device=device, seed=0, env_offset=offset)
            gen = torch.Generator(device=device).manual_seed(1000 + rank)
            tape = torch.randint(4, (blocks, chunk, N), generator=gen, device=device, dtype=torch.int64)

        def launch(i):
            if args.dry_run:
                time.sleep(0.001)
                return None
            return env.rollout(tape[i % blocks])

        for i in range(W):
            launch(i)
        sync()
        barrier()
        sync()
        events = []
        t0 = time.perf_counter()
        for i in range(W, W + K):
            if args.dry_run:
                launch(i)
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(i)
            e1.record()
            events.append((e0, e1))
        sync()
        barrier()
        sync()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed, float(N * chunk * K)], dtype=torch.float64, device=device)
        if distributed:
            dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        avg = sum(e0.elapsed_time(e1) for e0, e1 in events) * 1e-3 / len(events) if events else float(t[0].item()) / K
        del env, tape
        return float(t[0].item()), float(t[1].item()), avg, blocks

    elapsed, total_env_steps, avg_launch_s, blocks = timed_region(N, offset, chunk, K, W)

    # BASELINE configs[2] split over the ranks, same protocol (see the module docstring)
    g3 = 65536
    off3, n3 = shard_range(g3, rank, n_gpus)
    chunk3 = 64 if n3 > 16384 else 128
    k3, w3 = (K, W) if args.workload == 'cfg3' else (min(K, 20), 3)
    if args.workload == 'cfg3' and (args.num_envs in (None, g3)) and args.chunk in (None, chunk3):
        el3, tot3, avg3 = elapsed, total_env_steps, avg_launch_s
    else:
        el3, tot3, avg3, _ = timed_region(n3, off3, chunk3, k3, w3)

    rc = 0
    if rank == 0:
        per = algorithmic_bytes_per_env_step(SIZE, OBS_ELEMS)
        bytes_per_launch = per * N * chunk
        achieved = bytes_per_launch / avg_launch_s / 1e9
        traffic, frac_real = None, None
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath) and not args.dry_run:
            try:  # counter bytes of exactly this launch shape (N, S, chunk), else null
                traffic = json.load(open(tpath)).get(f'rollout_{N}x{SIZE}_chunk{chunk}')
            except Exception:
                traffic = None
        if traffic:
            frac_real = traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS
        lane = N >= LANE_ROLLOUT_MIN_ENVS
        kernel = ('wurm::lane_rollout_kernel<EPW, partial> (9x9 SingleSnake, partial_2 crop, RNG mode; one env per LANE, '
                  'wurm_amd/csrc/lane_rollout.hpp)') if lane else \
            'wurm::rollout_s9_kernel<4> (9x9 SingleSnake, partial_n crop, RNG mode; one wave per env)'
        line = {
            'metric': 'env_steps_per_s', 'value': total_env_steps / elapsed, 'unit': 'env-steps/s',
            'n_gpus': n_gpus, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': 'i32',
            'data': 'synthetic',
            'config': {'workload': f'SingleSnake {N} envs/GPU ({int(global_envs)} in total) size={SIZE} obs={OBS_MODE} '
                                   f'random actions; one bench step = one fused rollout launch of {chunk} batch-steps '
                                   f'(step+observe+reset per batch-step) = {N * chunk} env-steps per GPU; {K} such '
                                   f'launches timed after {W} untimed',
                       'baseline_config': 'BASELINE.json configs[1]' if args.workload == 'cfg2'
                                          else 'BASELINE.json configs[2]',
                       'num_envs_per_gpu': N, 'global_num_envs': int(global_envs), 'size': SIZE,
                       'observation_mode': OBS_MODE, 'batch_steps_per_launch': chunk,
                       'env_steps_per_launch_per_gpu': N * chunk,
                       'parallelism': f'env-batch split x{n_gpus} (no data-path collective)',
                       'world_size': world, 'backend': (dist.get_backend() if distributed else None),
                       'state_dtype': 'fp32 NCHW (exact integers)',
                       'action_tape_blocks': blocks},
            # cfg2 (512 lone waves on 1024 SIMDs) is bound by how fast ONE wave issues instructions, not by HBM: `frac` is
            # a statement about the SURVEY byte model of an unfused step / reset pair, `frac_real` about HBM utilisation
            'roofline': {'bound': 'hbm' if lane else 'issue', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'peak_achievable': HBM_ACHIEVABLE_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'frac_real': frac_real,
                         'kernel': kernel, 'avg_launch_ms': avg_launch_s * 1e3,
                         'algorithmic_bytes_per_env_step': per, 'env_steps_per_launch': N * chunk,
                         'limiter': 'observation stores (HBM)' if lane else 'instruction issue of one wave per env',
                         'note': '`achieved` prices the launch at the SURVEY §8(d) bytes of an UNFUSED step/reset pair; '
                                 '`traffic` (rocprofv3 FETCH_SIZE + WRITE_SIZE per launch of this exact shape, '
                                 'profiles/hbm_traffic.json) and `frac_real` are what the fused kernel really moves — '
                                 'far less, because the env state never leaves the chip between steps. ' +
                                 ('Batches of 6 144 envs and more run one env per lane and are bound by the observation '
                                  'store stream (DESIGN.md §4.9).' if lane else
                                  '512 envs are 512 lone waves on 1024 SIMDs: the kernel is bound by one wave\'s '
                                  'instruction issue rate, not by HBM (DESIGN.md §4.4).')},
            'cfg3_strong_scaling': {
                'value': tot3 / el3, 'unit': 'env-steps/s', 'scaling': 'strong', 'global_num_envs': g3,
                'num_envs_per_gpu': n3, 'batch_steps_per_launch': chunk3, 'steps': k3, 'warmup': w3,
                'ms_per_step': el3 / k3 * 1e3, 'avg_launch_ms': avg3 * 1e3,
                'obs_and_outputs_GBs_this_rank': (4 * OBS_ELEMS + 23) * n3 * chunk3 / avg3 / 1e9,
                'what': 'BASELINE configs[2]: SingleSnake 65 536 x 9 x 9 partial_2 split over the ranks (shard_range, no '
                        'data-path collective), same barrier + max-over-ranks protocol as the headline'},
        }
        if args.dry_run:
            line['dry_run'] = True
        if n_gpus == 1 and not args.no_extra and not args.dry_run:
            line['extra'] = extra_measurements(device)
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(512,
                                                    budget_s=0.5 if args.dry_run else 5.0)
            except Exception as e:  # the baseline is a reported extra: never lose the bench line over it
                line['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(line))
        sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def _timed(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def extra_measurements(device):
    """Secondary numbers (not the headline): the per-call Python API on the same workload, the other BASELINE configs
    through the fused rollout, the fused acting loop.  GB/s figures are REAL streams (the observation bytes the launch
    writes), never the unfused byte model."""
    import torch
    from wurm_amd.envs import SingleSnake, MultiSnake
    out = {}
    # (a) step()/reset() call pairs from Python — the drop-in loop of experiments/main.py:212-227
    N, T = 512, 4000
    env = SingleSnake(num_envs=N, size=SIZE, observation_mode=OBS_MODE, device=device, seed=0)
    actions = torch.randint(4, (T + 400, N), device=device, dtype=torch.int64)
    # `per_call_api_512` is the reference's own call form (SingleSnake.reset always returns its observation,
    # single_snake.py:322-342; experiments/main.py:212-227 discards it); the keyword form is this build's extension
    for variant, kw in (('per_call_api_512', {}), ('per_call_api_512_no_reset_obs', {'return_observations': False})):
        for t in range(400):
            _, _, d, _ = env.step(actions[t])
            env.reset(d, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(400, 400 + T):
            _, _, d, _ = env.step(actions[t])
            env.reset(d, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[variant] = {'value': N * T / dt, 'unit': 'env-steps/s', 'us_per_batch_step': dt / T * 1e6,
                        'what': 'Python loop of `obs, r, d, info = env.step(a); env.reset(d%s)`'
                                % (', return_observations=False' if kw else '')}
    del env, actions

    # (a') the same Python loop at BASELINE configs[2] whole (65 536 envs on one GPU: lane_step_kernel) and configs[3]
    def per_call_case(key, env, step_args, reset_arg, T, what, reset_kw={'return_observations': False}, check=False):
        def it(t):
            out_ = env.step(step_args(t))
            env.reset(reset_arg(out_[2]), **reset_kw)
            if check:
                env.check_consistency()
        for t in range(10):
            it(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(10, 10 + T):
            it(t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[key] = {'value': env.num_envs * T / dt, 'unit': 'env-steps/s', 'us_per_batch_step': dt / T * 1e6, 'what': what}
        t = traffic_detail.get({'per_call_api_cfg3_65536': 'resident_step_65536x9_partial2_reset_obs',
                                'per_call_api_cfg3_65536_no_reset_obs': 'resident_step_65536x9_partial2',
                                'per_call_api_cfg5_8192x36_default': 'grid_step_8192x36_default',
                                'per_call_api_cfg4_4096x25_k4': 'multi_step_cfg4_4096x25_k4_full',
                                'per_call_api_cfg4_4096x25_k4_no_mirror': 'multi_step_cfg4_4096x25_k4_full_no_mirror',
                                'per_call_api_cfg5_8192x36_default_no_mirror': 'grid_step_8192x36_default_no_mirror'}.get(key, key))
        if t:  # rocprofv3 FETCH_SIZE + WRITE_SIZE of one iteration's launches (profiles/hbm_traffic.json)
            out[key]['traffic_bytes_per_batch_step'] = t['total_bytes']
            out[key]['frac_real'] = t['total_bytes'] / (dt / T) / 1e9 / HBM_PEAK_GBS

    try:
        traffic_detail = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'))).get('detail', {})
    except Exception:
        traffic_detail = {}

    # BASELINE configs[2] whole on one GPU.  As for 512 envs, the plain key is the reference's own call form (reset(d) returns
    # its observation); from 4096 envs of 9 x 9 the step runs on the resident mirror of the state
    # (wurm_amd/csrc/lane_resident.hpp) — `_no_mirror` switches it off (lane_step_kernel, what rounds 2 measured)
    N, T = 65536, 200
    acts = torch.randint(4, (T + 10, N), device=device, dtype=torch.int64)
    per_call_case('per_call_api_cfg3_65536', SingleSnake(N, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
                  lambda t: acts[t], lambda d: d, T,
                  'BASELINE configs[2] whole on one GPU through `env.step(a); env.reset(d)` (resident mirror, lazy)', reset_kw={})
    per_call_case('per_call_api_cfg3_65536_no_reset_obs', SingleSnake(N, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
                  lambda t: acts[t], lambda d: d, T,
                  'the same through `env.step(a); env.reset(d, return_observations=False)`')
    os.environ['WURM_RESIDENT_MIN_ENVS'] = str(10 ** 9)
    try:
        per_call_case('per_call_api_cfg3_65536_no_mirror', SingleSnake(N, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
                      lambda t: acts[t], lambda d: d, T,
                      '`env.step(a); env.reset(d)` with the mirror switched off (WURM_RESIDENT_MIN_ENVS): lane_step_kernel '
                      'reads the whole (N,3,9,9) state every call', reset_kw={})
    finally:
        os.environ.pop('WURM_RESIDENT_MIN_ENVS')
    # algorithmic bytes of one resident step launch: two crops of 300 B per env, the mirror read and written (2 x 32 B),
    # action in and out (16 B), reward + 4 flag bytes + the postponed reset's flag (9 B)
    for key, per_env in (('per_call_api_cfg3_65536', 600 + 64 + 16 + 9), ('per_call_api_cfg3_65536_no_reset_obs', 300 + 64 + 16 + 9)):
        e = out[key]
        e['algorithmic_bytes_per_batch_step'] = per_env * N
        e['frac_algorithmic'] = per_env * N / (e['us_per_batch_step'] * 1e-6) / 1e9 / HBM_PEAK_GBS
    N, T = 8192, 200
    acts = torch.randint(4, (T + 10, N), device=device, dtype=torch.int64)
    per_call_case('per_call_api_cfg5_8192x36_default', SingleSnake(N, 36, observation_mode='default', device=device, seed=0),
                  lambda t: acts[t], lambda d: d, T,
                  'BASELINE configs[4] through `env.step(a); env.reset(d, return_observations=False)` (clock grids kept in the '
                  'resident mirror, lazy: grid_rollout.hip)')
    os.environ['WURM_RESIDENT_MIN_ENVS'] = str(10 ** 9)
    try:
        per_call_case('per_call_api_cfg5_8192x36_default_no_mirror',
                      SingleSnake(N, 36, observation_mode='default', device=device, seed=0), lambda t: acts[t], lambda d: d, T,
                      'the same with the mirror switched off: the step reads the (N,3,36,36) fp32 state every call')
    finally:
        os.environ.pop('WURM_RESIDENT_MIN_ENVS')
    from wurm_amd.envs import SimpleGridworld
    N, T = 64, 2000
    acts = torch.randint(4, (T + 10, N), device=device, dtype=torch.int64)
    per_call_case('per_call_api_cfg1_gridworld_64x9',
                  SimpleGridworld(N, 9, start_location=(4, 4), observation_mode='default', device=device, seed=0),
                  lambda t: acts[t], lambda d: d, T,
                  'BASELINE configs[0] (SimpleGridworld 64 x 9 x 9, default observation) on the GPU, same loop')
    N, K, T = 4096, 4, 100
    acts = torch.randint(8, (T + 10, K, N), device=device, dtype=torch.int64)
    keys = [f'agent_{i}' for i in range(K)]
    per_call_case('per_call_api_cfg4_4096x25_k4', MultiSnake(N, K, 25, device=device, seed=0),
                  lambda t: dict(zip(keys, acts[t].unbind(0))), lambda d: d['__all__'], T,
                  "BASELINE configs[3] through `env.step(actions); env.reset(dones['__all__'], return_observations=False)` "
                  '(grids kept in the resident mirror, lazy: multi_snake.hip)')
    os.environ['WURM_RESIDENT_MIN_ENVS'] = str(10 ** 9)
    try:
        per_call_case('per_call_api_cfg4_4096x25_k4_no_mirror', MultiSnake(N, K, 25, device=device, seed=0),
                      lambda t: dict(zip(keys, acts[t].unbind(0))), lambda d: d['__all__'], T,
                      'the same with the mirror switched off: the step reads foods / heads / bodies (92 MB of fp32) every call')
    finally:
        os.environ.pop('WURM_RESIDENT_MIN_ENVS')
    # (a'') the MultiSnake variants the reference itself runs: tests/test_multi_snake_env.py:100-104 (training dynamics,
    # `partial_5` crops; SURVEY §8(d) "additionally") and experiments/speeds.py:10-44 (10 agents on 36 x 36, respawn 'any',
    # `step; reset(done['__all__']); check_consistency()` — the reference's own benchmark loop, reset observation included)
    def cfg4prime():
        return MultiSnake(4096, 4, 25, device=device, seed=0, respawn_mode='any', food_mode='random_rate',
                          boost_cost_prob=0.25, observation_mode='partial_5', food_on_death_prob=0.33, food_rate=2.5e-4)

    def speeds_env():
        return MultiSnake(4096, 10, 36, device=device, seed=0, boost=True, respawn_mode='any')
    per_call_case('per_call_api_cfg4prime_4096x25_k4_partial5', cfg4prime(),
                  lambda t: dict(zip(keys, acts[t].unbind(0))), lambda d: d['__all__'], T,
                  "MultiSnake 4096x25x25 K=4 with the reference's training dynamics (tests/test_multi_snake_env.py:100-104: "
                  "respawn 'any', random_rate food, partial_5), `env.step(a); env.reset(d['__all__'], return_observations=False)`")
    N, K, T = 4096, 10, 30
    acts = torch.randint(8, (T + 10, K, N), device=device, dtype=torch.int64)
    keys10 = [f'agent_{i}' for i in range(K)]
    per_call_case('per_call_api_speeds_4096x36_k10', speeds_env(),
                  lambda t: dict(zip(keys10, acts[t].unbind(0))), lambda d: d['__all__'], T,
                  "experiments/speeds.py shape (4096 x 36 x 36, 10 agents), `step; reset(d['__all__'], return_observations=False)`")
    per_call_case('speeds_py_loop_4096x36_k10', speeds_env(),
                  lambda t: dict(zip(keys10, acts[t].unbind(0))), lambda d: d['__all__'], T,
                  "experiments/speeds.py:30-38 as written: `step(actions); reset(done['__all__']); check_consistency()` "
                  '(the reset returns its 10 observations, the checker runs every step)', reset_kw={}, check=True)
    del acts

    def rollout_case(key, make_env, shape_actions, A, chunk, reps, obs_bytes, what, traffic_key=None):
        env = make_env()
        acts = torch.randint(A, (reps + 1,) + shape_actions(chunk), device=device, dtype=torch.int64)
        it = iter(range(reps + 1))
        dt = _timed(lambda: env.rollout(acts[next(it)]), reps)
        n_env = env.num_envs
        eps = n_env * chunk / dt
        out[key] = {'value': eps, 'unit': 'env-steps/s', 'ms_per_launch': dt * 1e3, 'batch_steps_per_launch': chunk,
                    'obs_write_GBs': obs_bytes * eps / 1e9, 'obs_write_frac_of_hbm_peak': obs_bytes * eps / 1e9 / HBM_PEAK_GBS,
                    'what': what}
        t = traffic_detail.get(traffic_key) if traffic_key else None
        if t:  # rocprofv3 FETCH_SIZE + WRITE_SIZE of exactly this launch shape (profiles/hbm_traffic.json)
            out[key]['traffic_bytes_per_launch'] = t['total_bytes']
            out[key]['frac_real'] = t['total_bytes'] / dt / 1e9 / HBM_PEAK_GBS

    # (b) one GPU's share of configs[2] (8192 envs) and all of configs[2] on one GPU
    rollout_case('rollout_8192', lambda: SingleSnake(8192, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
                 lambda c: (c, 8192), 4, 128, 16, 4 * OBS_ELEMS + 7 + 16,
                 'BASELINE configs[2] per-GPU share (65536/8), fused rollout, 128 batch-steps per launch')
    rollout_case('rollout_cfg3_65536', lambda: SingleSnake(65536, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
                 lambda c: (c, 65536), 4, 64, 8, 4 * OBS_ELEMS + 7 + 16,
                 'BASELINE configs[2] whole (65536 envs) on ONE GPU, fused rollout, 64 batch-steps per launch')
    # (c) BASELINE configs[4]: SingleSnake 8192 x 36 x 36, default RGB observation
    rollout_case('rollout_cfg5_8192x36_default',
                 lambda: SingleSnake(8192, 36, observation_mode='default', device=device, seed=0),
                 lambda c: (c, 8192), 4, 16, 6, 12 * 36 * 36,
                 'BASELINE configs[4]: SingleSnake 8192x36x36 default-RGB obs, fused rollout, 16 batch-steps per launch '
                 '(the launch writes the 15 552 B observation per env-step; SURVEY byte model of an unfused pair: 41 511 B)',
                 traffic_key='rollout_cfg5_8192x36_default_chunk16')
    rollout_case('rollout_cfg5_8192x36_default_64steps',
                 lambda: SingleSnake(8192, 36, observation_mode='default', device=device, seed=0),
                 lambda c: (c, 8192), 4, 64, 4, 12 * 36 * 36,
                 'BASELINE configs[4] as above with 64 batch-steps per launch (the state load / store and the launch are '
                 'amortised over four times as many steps)')
    # (d) BASELINE configs[3]: MultiSnake 4096 x 25 x 25, 4 agents, constructor defaults ('full' obs)
    rollout_case('multi_rollout_cfg4_4096x25_k4_full_64steps', lambda: MultiSnake(4096, 4, 25, device=device, seed=0),
                 lambda c: (c, 4, 4096), 8, 64, 4, 12 * 4 * 25 * 25,
                 'BASELINE configs[3] with 64 batch-steps per launch')
    rollout_case('multi_rollout_cfg4_4096x25_k4_full', lambda: MultiSnake(4096, 4, 25, device=device, seed=0),
                 lambda c: (c, 4, 4096), 8, 16, 6, 12 * 4 * 25 * 25,
                 'BASELINE configs[3]: MultiSnake 4096x25x25, 4 agents, defaults, step+observe+reset(__all__) per '
                 'batch-step, fused rollout, 16 batch-steps per launch (30 000 B of observations per env-step; SURVEY byte '
                 'model of an unfused pair: 75 160 B; reference torch-CPU: 3 280 env-steps/s)',
                 traffic_key='multi_rollout_cfg4_4096x25_k4_full_chunk16')
    rollout_case('multi_rollout_cfg4prime_4096x25_k4_partial5', cfg4prime, lambda c: (c, 4, 4096), 8, 16, 6, 12 * 4 * 11 * 11,
                 "MultiSnake 4096x25x25 K=4, training dynamics (respawn 'any', random_rate food) and partial_5 crops "
                 '(tests/test_multi_snake_env.py:100-104), fused rollout, 16 batch-steps per launch',
                 traffic_key='multi_rollout_cfg4prime_4096x25_k4_partial5_chunk16')
    rollout_case('multi_rollout_speeds_4096x36_k10', speeds_env, lambda c: (c, 10, 4096), 8, 4, 4, 12 * 10 * 36 * 36,
                 "experiments/speeds.py shape: MultiSnake 4096x36x36, 10 agents, respawn 'any', 'full' observations "
                 '(155 520 B per env-step), fused rollout, 4 batch-steps per launch',
                 traffic_key='multi_rollout_speeds_4096x36_k10_chunk4')
    # (e) the acting loop with the policy inside the env kernel: MLP 75->64->64->{4,1} + sampling
    from wurm_amd.agents import FeedforwardAgent, pack_policy_params
    N, T, reps = 512, 256, 8
    torch.manual_seed(0)
    env = SingleSnake(num_envs=N, size=SIZE, observation_mode=OBS_MODE, device=device, seed=0)
    params = pack_policy_params(FeedforwardAgent(4, 2, 64, OBS_ELEMS).to(device))
    state = [env.reset()]

    def act():
        state[0] = env.policy_rollout(params, state[0], T, check=False)['state']
    dt = _timed(act, reps)
    out['policy_rollout_512'] = {
        'value': N * T / dt, 'unit': 'env-steps/s',
        'what': 'policy forward (random-init FeedforwardAgent) + Categorical sample + step + observe + reset per '
                'env-step, fused in one kernel, launches of 256 batch-steps'}
    return out


def main() -> int:
    args = parse_args()
    in_world = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if args.gpus > 1 and not in_world and not args.worker:
        return spawn_ranks(args)
    return worker(args)


if __name__ == '__main__':
    sys.exit(main())
