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
    t_init = 0.0
    if distributed:
        import torch.distributed as dist
        t_i0 = time.perf_counter()
        if args.dry_run:
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            dist.barrier()   # (RCCL builds its communicator on first use: counted as init, not as bench time)
        t_init = time.perf_counter() - t_i0
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
        # ONE pair of HIP events around the K launches, on the stream they are issued on: an event pair per launch (rounds 2-5)
        # put two timestamp packets between consecutive kernels — 10 us during which the GPU ran nothing (rocprofv3
        # --kernel-trace: end-to-start gap 10.3 us with them, 0 without), i.e. 4 % of the headline it was there to price
        ev = None if args.dry_run else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        t0 = time.perf_counter()
        if ev:
            ev[0].record()
        for i in range(W, W + K):
            launch(i)
        if ev:
            ev[1].record()
        sync()
        barrier()
        sync()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed, float(N * chunk * K)], dtype=torch.float64, device=device)
        if distributed:
            dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        avg = ev[0].elapsed_time(ev[1]) * 1e-3 / K if ev else float(t[0].item()) / K
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

    # every rank's own record: the first run on a multi-GPU node should explain itself
    mine = {'rank': rank, 'local_rank': local_rank, 'device': None if args.dry_run else torch.cuda.get_device_name(local_rank),
            'env_offset': int(offset), 'num_envs': int(N), 'cfg3_shard': [int(off3), int(off3 + n3)],
            'comm_init_s': round(t_init, 3), 'avg_launch_ms': round(avg_launch_s * 1e3, 4),
            'cfg3_avg_launch_ms': round(avg3 * 1e3, 4), 'host': os.uname().nodename, 'pid': os.getpid()}
    ranks = [mine]
    if distributed:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)

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
                         'note': '`achieved` = SURVEY 8(d) bytes of an UNFUSED step/reset pair; `traffic` / `frac_real` = '
                                 'rocprofv3 counter bytes of this launch shape (profiles/hbm_traffic.json)'},
            'cfg3_strong_scaling': {
                'value': tot3 / el3, 'unit': 'env-steps/s', 'scaling': 'strong', 'global_num_envs': g3,
                'num_envs_per_gpu': n3, 'batch_steps_per_launch': chunk3, 'steps': k3, 'warmup': w3,
                'ms_per_step': el3 / k3 * 1e3, 'avg_launch_ms': avg3 * 1e3,
                'obs_and_outputs_GBs_this_rank': (4 * OBS_ELEMS + 23) * n3 * chunk3 / avg3 / 1e9,
                'what': 'BASELINE configs[2] split over the ranks, same protocol as the headline'},
        }
        line['ranks'] = ranks
        if args.dry_run:
            line['dry_run'] = True
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(512,
                                                    budget_s=0.5 if args.dry_run else 5.0)
            except Exception as e:  # the baseline is a reported extra: never lose the bench line over it
                line['cpu_baseline'] = {'error': repr(e)}
        if n_gpus == 1 and not args.no_extra and not args.dry_run:
            line['extra'] = extra_measurements(device)
            for k, what in DESCRIPTIONS.items():   # what each extra measures: stderr (and profiles/bench_keys.md), not the line
                print(f'# {k}: {what}', file=sys.stderr)
            try:
                os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                with open(os.path.join(ROOT, 'gpurun_out', 'bench_keys.md'), 'w') as f:
                    f.write('# bench.py extras: what each key measures\n\n' +
                            ''.join(f'* `{k}` — {what}\n' for k, what in DESCRIPTIONS.items()))
            except OSError:
                pass
        line['key'] = key_numbers(line)   # LAST: the driver keeps the tail of the line
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


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


REPEATS = 3          # every extra is timed this many times; the record carries the median and the minimum
DESCRIPTIONS = {}    # key -> what it measures (printed to stderr, kept in profiles/bench_keys.md; not in the JSON line)


def _describe(key, what):
    DESCRIPTIONS[key] = what


def host_calibration(device):
    """Host-side unit costs on this box (the small-batch per-call numbers are HOST-bound: VERDICT r03 saw cfg1 at 24 us per
    iteration on the driver box against 7-10 us elsewhere — these figures say how fast the box's host side is)."""
    import torch
    from wurm_amd import _lib
    l = _lib.lib()
    out = {}
    n = 20000
    t0 = time.perf_counter()
    for _ in range(n):
        l.wurm_launch_count()
    out['ctypes_call_ns'] = (time.perf_counter() - t0) / n * 1e9
    t0 = time.perf_counter()
    for _ in range(200000):
        pass
    out['py_loop_iter_ns'] = (time.perf_counter() - t0) / 200000 * 1e9
    x = torch.zeros(64, device=device)
    for _ in range(300):   # (warm: the first calibration of a process measured 37 us against 3.4 us afterwards — VERDICT r04)
        x.add_(1.0)
        torch.empty(64, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        x.add_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    out['torch_small_kernel_issue_us'] = (t1 - t0) / 2000 * 1e6
    out['torch_small_kernel_drain_us'] = (time.perf_counter() - t0) / 2000 * 1e6
    t0 = time.perf_counter()
    for _ in range(2000):
        torch.empty(64, device=device)
    out['torch_empty_us'] = (time.perf_counter() - t0) / 2000 * 1e6
    # what THIS box's HBM takes from a plain kernel: a 2 GB fill (the store ceiling of the observation streams: boxes of the
    # pool differ by +-10 %) and a 1 GB copy
    big = torch.empty(1 << 29, dtype=torch.float32, device=device)
    big.fill_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        big.fill_(2.0)
    torch.cuda.synchronize()
    out['hbm_fill_2GB_TBps'] = 5 * big.numel() * 4 / (time.perf_counter() - t0) / 1e12
    half = big.numel() // 2
    big[half:].copy_(big[:half])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        big[half:].copy_(big[:half])
    torch.cuda.synchronize()
    out['hbm_copy_1GB_read_plus_write_TBps'] = 5 * 2 * half * 4 / (time.perf_counter() - t0) / 1e12
    del big
    return {k: round(v, 2) for k, v in out.items()}


def extra_measurements(device):
    """Secondary numbers (not the headline): the per-call Python API, the other BASELINE configs through the fused rollout,
    the loops the reference ships.  Every entry: median and minimum of REPEATS timings.  GB/s figures are REAL streams
    (rocprofv3 counter bytes of the same launch shape, profiles/hbm_traffic.json), never the unfused byte model.
    The most important entries come LAST (the driver keeps the tail of the line)."""
    import torch
    from wurm_amd import _lib
    from wurm_amd.envs import SingleSnake, MultiSnake, SimpleGridworld
    from wurm_amd.utils import env_consistency
    out = {}
    l = _lib.lib()
    try:
        traffic_detail = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'))).get('detail', {})
    except Exception:
        traffic_detail = {}

    def machine(env):
        fs = getattr(env, '_fs', None)
        if fs is not None:
            m = 'C Stepper' if type(fs).__name__ == 'Stepper' else 'PyStepper'
        else:
            fn = getattr(env, '_mc_fn', None)
            m = 'C shim' if fn is not None and 'partial' in type(fn).__name__ else 'ctypes'
        ms = env.mirror_state()
        return m + ('+torchinfo' if _lib.torch_helpers() else '') + ', mirror ' + ms['state']

    def per_call(key, make_env, step_args, reset_arg, T, what, reset_kw=None, after_step=None, traffic_key=None):
        """T iterations of `out = env.step(a); [after_step(env, out)]; env.reset(done, **reset_kw)`, REPEATS times"""
        _describe(key, what)
        reset_kw = {'return_observations': False} if reset_kw is None else reset_kw
        env = make_env()

        def it(t):
            o = env.step(step_args(t))
            if after_step is not None:
                after_step(env, o)
            env.reset(reset_arg(o[2]), **reset_kw)
        for t in range(10):
            it(t)
        times, launches = [], 0
        for r in range(REPEATS):
            torch.cuda.synchronize()
            c0 = l.wurm_launch_count()
            t0 = time.perf_counter()
            for t in range(10, 10 + T):
                it(t)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / T)
            launches = (l.wurm_launch_count() - c0) / T
        med = _median(times)
        e = out[key] = {'value': env.num_envs * 1.0 / med, 'us': round(med * 1e6, 2), 'us_min': round(min(times) * 1e6, 2),
                        'launches_per_iter': round(launches, 2), 'machine': machine(env)}
        t = traffic_detail.get(traffic_key or key)
        if t:  # rocprofv3 FETCH_SIZE + WRITE_SIZE of one iteration's launches (profiles/hbm_traffic.json)
            e['traffic_bytes'] = round(t['total_bytes'])
            e['frac_real'] = round(t['total_bytes'] / med / 1e9 / HBM_PEAK_GBS, 4)
        return e

    def rollout(key, make_env, shape_actions, A, chunk, reps, what, traffic_key=None):
        _describe(key, what)
        env = make_env()
        acts = torch.randint(A, (reps + 1,) + shape_actions(chunk), device=device, dtype=torch.int64)
        times = []
        for r in range(REPEATS):
            it = iter(range(reps + 1))
            times.append(_timed(lambda: env.rollout(acts[next(it)]), reps))
        med = _median(times)
        e = out[key] = {'value': env.num_envs * chunk / med, 'ms': round(med * 1e3, 4), 'ms_min': round(min(times) * 1e3, 4),
                        'batch_steps_per_launch': chunk}
        t = traffic_detail.get(traffic_key) if traffic_key else None
        if t:  # counter bytes of exactly this launch shape
            e['traffic_bytes'] = round(t['total_bytes'])
            e['frac_real'] = round(t['total_bytes'] / med / 1e9 / HBM_PEAK_GBS, 4)
            e['frac_real_best'] = round(t['total_bytes'] / min(times) / 1e9 / HBM_PEAK_GBS, 4)
        return e

    def guarded(fn):   # an extra is an extra: a failing case leaves {'error': ...} under its key, never loses the line
        def run(key, *a, **kw):
            try:
                return fn(key, *a, **kw)
            except Exception as e:
                out[key] = {'error': repr(e)[:300]}
                torch.cuda.synchronize()
                return out[key]
        return run
    def alloc_spread(key, make_env, shape_actions, A, chunk, n_blocks, what):
        """The HBM-bound rollouts take 0.37-0.48 ms for the SAME launch depending on the allocation their output lies in
        (profiles/r04_placement_probe.txt): p10 / p50 / p90 of one launch over `n_blocks` fresh output allocations of this
        process (earlier outputs are kept alive so that every launch gets a block of its own; each block is written once
        untimed first)."""
        _describe(key, what)
        env = make_env()
        acts = torch.randint(A, (2,) + shape_actions(chunk), device=device, dtype=torch.int64)
        held, times = [], []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for b in range(n_blocks):
            o = env.rollout(acts[0])
            del o                       # (the allocator hands the block that was just freed to the next launch)
            torch.cuda.synchronize()
            e0.record()
            o = env.rollout(acts[1])
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
            held.append(o)
        times.sort()
        q = lambda f: times[min(len(times) - 1, int(f * len(times)))]   # noqa: E731
        out[key] = {'value': env.num_envs * chunk / (q(0.5) * 1e-3), 'ms_p10': round(q(0.1), 4), 'ms_p50': round(q(0.5), 4),
                    'ms_p90': round(q(0.9), 4), 'ms_min': round(times[0], 4), 'ms_max': round(times[-1], 4),
                    'allocations': n_blocks, 'batch_steps_per_launch': chunk}
        del held, o
        torch.cuda.empty_cache()        # (the blocks go back to the driver: nothing measured later sees this extra's allocations)
        return out[key]

    per_call, rollout, alloc_spread = guarded(per_call), guarded(rollout), guarded(alloc_spread)

    out['host_calibration_before'] = host_calibration(device)
    _describe('host_calibration_*', 'host unit costs of this box: a ctypes call, a Python loop iteration, issuing / draining a '
              'tiny torch kernel, torch.empty — before and after the extras')

    # ---- MultiSnake variants the reference itself runs (not BASELINE configs)
    def cfg4prime(**kw):
        return MultiSnake(4096, 4, 25, device=device, seed=0, respawn_mode='any', food_mode='random_rate',
                          boost_cost_prob=0.25, observation_mode='partial_5', food_on_death_prob=0.33, food_rate=2.5e-4, **kw)

    def speeds_env(**kw):
        return MultiSnake(4096, 10, 36, device=device, seed=0, boost=True, respawn_mode='any', **kw)
    N, K, T = 4096, 4, 60
    T4 = T
    acts4 = torch.randint(8, (T4 + 10, K, N), device=device, dtype=torch.int64)
    keys4 = [f'agent_{i}' for i in range(K)]
    # (the action dicts are what the policy hands over: built before the timed loops — indexing the tape and unbinding it
    # inside them cost ~5 us per iteration of benchmark, not of env)
    dicts4 = [dict(zip(keys4, acts4[t].unbind(0))) for t in range(T4 + 10)]
    a4 = lambda t: dicts4[t]   # noqa: E731
    d_all = lambda d: d['__all__']                        # noqa: E731
    per_call('per_call_cfg4prime_partial5', cfg4prime, a4, d_all, T,
             "MultiSnake 4096x25x25 K=4, the reference's training dynamics (tests/test_multi_snake_env.py:100-104: respawn "
             "'any', random_rate food, partial_5): `step(a); reset(d['__all__'], return_observations=False)`",
             traffic_key='per_call_api_cfg4prime_4096x25_k4_partial5')
    # (the dicts above are rows of ONE (K, N) tape — the zero-copy case of Stepper.step_multi.  A policy that emits
    # one tensor per agent, as experiments/multiagent.py does, hands over K separately allocated tensors: the class stacks them)
    split4 = [{k: v.clone() for k, v in d.items()} for d in dicts4]
    per_call('per_call_cfg4prime_partial5_split_actions', cfg4prime, lambda t: split4[t], d_all, T,
             "per_call_cfg4prime_partial5 with K separately allocated action tensors per step (stacked by the class; the "
             "other per_call_multi_* / cfg4* keys feed rows of one (K, N) tape, the zero-copy case)")
    del split4
    acts10 = torch.randint(8, (40, 10, N), device=device, dtype=torch.int64)
    keys10 = [f'agent_{i}' for i in range(10)]
    dicts10 = [dict(zip(keys10, acts10[t].unbind(0))) for t in range(40)]
    a10 = lambda t: dicts10[t]   # noqa: E731
    per_call('per_call_speeds_4096x36_k10', speeds_env, a10, d_all, 30,
             "experiments/speeds.py shape (4096 x 36 x 36, 10 agents): `step; reset(d['__all__'], return_observations=False)`",
             traffic_key='per_call_api_speeds_4096x36_k10')
    per_call('speeds_py_loop_4096x36_k10', speeds_env, a10, d_all, 30,
             "experiments/speeds.py:30-38 as written: `step(actions); reset(done['__all__']); check_consistency()`",
             reset_kw={}, after_step=None)
    # (check_consistency() after the reset, as speeds.py has it)
    _describe('speeds_py_loop_check_4096x36_k10', 'the same loop with env.check_consistency() after the reset (speeds.py:37)')
    env = speeds_env()
    for t in range(5):
        o = env.step(a10(t)); env.reset(o[2]['__all__']); env.check_consistency()
    times = []
    for r in range(REPEATS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(5, 30):
            o = env.step(a10(t)); env.reset(o[2]['__all__']); env.check_consistency()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 25)
    out['speeds_py_loop_check_4096x36_k10'] = {'value': 4096 / _median(times), 'us': round(_median(times) * 1e6, 1),
                                               'us_min': round(min(times) * 1e6, 1), 'machine': machine(env)}
    del env, acts10
    rollout('multi_rollout_cfg4prime_partial5', cfg4prime, lambda c: (c, 4, 4096), 8, 16, 6,
            "MultiSnake 4096x25x25 K=4, training dynamics + partial_5 crops, fused rollout, 16 batch-steps per launch",
            traffic_key='multi_rollout_cfg4prime_4096x25_k4_partial5_chunk16')
    rollout('multi_rollout_speeds_4096x36_k10', speeds_env, lambda c: (c, 10, 4096), 8, 4, 4,
            "experiments/speeds.py shape, 'full' observations (155 520 B per env-step), fused rollout, 4 batch-steps per launch",
            traffic_key='multi_rollout_speeds_4096x36_k10_chunk4')

    # ---- the acting loop with the policy inside the env kernel and the A2C loops the reference ships
    from wurm_amd.agents import FeedforwardAgent, pack_policy_params
    N, T, reps = 512, 256, 8
    torch.manual_seed(0)
    env = SingleSnake(num_envs=N, size=SIZE, observation_mode=OBS_MODE, device=device, seed=0)
    params = pack_policy_params(FeedforwardAgent(4, 2, 64, OBS_ELEMS).to(device))
    state = [env.reset()]

    def act():
        state[0] = env.policy_rollout(params, state[0], T, check=False)['state']
    times = [_timed(act, reps) for _ in range(REPEATS)]
    out['policy_rollout_512'] = {'value': N * T / _median(times), 'ms': round(_median(times) * 1e3, 4)}
    _describe('policy_rollout_512', 'policy forward (random-init FeedforwardAgent 75-64-64-{4,1}) + Categorical sample + step + '
              'observe + reset per env-step fused in one kernel, launches of 256 batch-steps')
    del env
    try:
        sys.path.insert(0, os.path.join(ROOT, 'examples'))
        import a2c_loop
        import a2c_fused_actor
        a2c_loop.run(steps=100, log_interval=100, verbose=False, device=device)
        h = [a2c_loop.run(steps=500, log_interval=500, verbose=False, device=device)[-1]['env_steps_per_s']
             for _ in range(REPEATS)]
        out['a2c_loop_512'] = {'value': _median(h), 'best': max(h), 'reference_readme': 4.2e4}
        _describe('a2c_loop_512', "examples/a2c_loop.py = the reference's single-agent experiment (experiments/main.py:194-247; "
                  "README.md:89-97: 512 envs, size 9, partial_2, feed-forward agent, update every 5 steps; the reference "
                  "quotes 10M env-steps in about four minutes = 4.2e4 env-steps/s on a GTX 1080 Ti), env-steps/s incl. "
                  "policy forward / backward / Adam in torch")
        a2c_fused_actor.run(steps=200, log_interval=200, verbose=False, device=device)
        h = [a2c_fused_actor.run(steps=2000, log_interval=2000, verbose=False, device=device)[-1]['env_steps_per_s']
             for _ in range(REPEATS)]
        out['a2c_fused_actor_512'] = {'value': _median(h), 'best': max(h)}
        _describe('a2c_fused_actor_512', 'examples/a2c_fused_actor.py: the same experiment with the acting half (policy + sample '
                  '+ step + reset) fused into the env kernel, learner in torch')
    except Exception as e:  # an extra: never lose the line over it
        out['a2c_loop_512'] = {'error': repr(e)[:200]}

    # ---- cfg5 and cfg1 per call, cfg5 rollouts
    N, T = 8192, 100
    acts = torch.randint(4, (T + 10, N), device=device, dtype=torch.int64)
    a1 = lambda t: acts[t]        # noqa: E731
    same = lambda d: d            # noqa: E731
    per_call('per_call_cfg5_8192x36_default_no_mirror',
             lambda: SingleSnake(N, 36, observation_mode='default', device=device, seed=0, resident_mirror=False), a1, same, T,
             'cfg5 per call with the mirror switched off: the step reads the (N,3,36,36) fp32 state every call',
             traffic_key='grid_step_8192x36_default_no_mirror')
    per_call('per_call_cfg5_8192x36_default', lambda: SingleSnake(N, 36, observation_mode='default', device=device, seed=0),
             a1, same, T, 'BASELINE configs[4] through `env.step(a); env.reset(d, return_observations=False)` (clock grids '
             'kept in the resident mirror, lazy: grid_rollout.hip)', traffic_key='grid_step_8192x36_default')
    rollout('rollout_cfg5_8192x36_default_64steps', lambda: SingleSnake(8192, 36, observation_mode='default', device=device, seed=0),
            lambda c: (c, 8192), 4, 64, 4, 'BASELINE configs[4] with 64 batch-steps per launch')
    rollout('rollout_cfg5_8192x36_default', lambda: SingleSnake(8192, 36, observation_mode='default', device=device, seed=0),
            lambda c: (c, 8192), 4, 16, 6, 'BASELINE configs[4]: SingleSnake 8192x36x36 default-RGB obs, fused rollout, 16 '
            'batch-steps per launch (15 552 B of observation per env-step)', traffic_key='rollout_cfg5_8192x36_default_chunk16')

    # ---- cfg3 (65 536 x 9 x 9) in the other observation modes (the reference's constructor default is one_channel)
    for mode in ('one_channel', 'default', 'raw', 'partial_3'):
        rollout(f'rollout_65536x9_{mode}', lambda: SingleSnake(65536, SIZE, observation_mode=mode, device=device, seed=0),
                lambda c: (c, 65536), 4, 32, 6, f'SingleSnake 65 536 x 9 x 9 with observation_mode={mode!r}, fused rollout, 32 '
                'batch-steps per launch', traffic_key=f'rollout_65536x9_{mode}_chunk32')

    # ---- the sizes between the 9 x 9 lane kernels and the clock grids (round 6: lane_wide.hpp, one env per lane on 128-bit masks)
    rollout('rollout_65536x10_partial_2', lambda: SingleSnake(65536, 10, observation_mode='partial_2', device=device, seed=0),
            lambda c: (c, 65536), 4, 32, 6, 'SingleSnake 65 536 x 10 x 10 partial_2, fused rollout, 32 batch-steps per launch',
            traffic_key='rollout_65536x10_partial_2_chunk32')
    rollout('rollout_65536x11_default', lambda: SingleSnake(65536, 11, observation_mode='default', device=device, seed=0),
            lambda c: (c, 65536), 4, 32, 6, 'SingleSnake 65 536 x 11 x 11 default (RGB), fused rollout, 32 batch-steps per launch',
            traffic_key='rollout_65536x11_default_chunk32')

    # ---- cfg4 (BASELINE configs[3]): MultiSnake 4096 x 25 x 25, 4 agents, constructor defaults ('full' observations)
    per_call('per_call_cfg4_no_mirror', lambda: MultiSnake(4096, 4, 25, device=device, seed=0, resident_mirror=False), a4, d_all, T4,
             'cfg4 per call with the mirror switched off: the step reads foods / heads / bodies (92 MB of fp32) every call',
             traffic_key='multi_step_cfg4_4096x25_k4_full_no_mirror')
    per_call('per_call_cfg4', lambda: MultiSnake(4096, 4, 25, device=device, seed=0), a4, d_all, T4,
             "BASELINE configs[3] through `env.step(actions); env.reset(dones['__all__'], return_observations=False)`",
             traffic_key='multi_step_cfg4_4096x25_k4_full')
    rollout('multi_rollout_cfg4_full_64steps', lambda: MultiSnake(4096, 4, 25, device=device, seed=0),
            lambda c: (c, 4, 4096), 8, 64, 4, 'BASELINE configs[3] with 64 batch-steps per launch')
    rollout('multi_rollout_cfg4_full', lambda: MultiSnake(4096, 4, 25, device=device, seed=0),
            lambda c: (c, 4, 4096), 8, 16, 6, 'BASELINE configs[3]: MultiSnake 4096x25x25, 4 agents, defaults, step + observe + '
            'reset(__all__) per batch-step, fused rollout, 16 batch-steps per launch (30 000 B of observations per env-step)',
            traffic_key='multi_rollout_cfg4_4096x25_k4_full_chunk16')
    del acts4, dicts4
    # small MultiSnake batches: host-bound (one C call per step: wurm_amd/csrc/fastcall.c Stepper.step_multi)
    for key, (n_, k_, s_), kw_, what in (
            ('per_call_multi_512x12_k2', (512, 2, 12), {}, "MultiSnake 512 x 12 x 12, 2 snakes, constructor defaults (the board size of "
             "the reference's own tests, tests/test_multi_snake_env.py:21-47)"),
            ('per_call_multi_512x25_k4_train_partial5', (512, 4, 25),
             dict(respawn_mode='any', food_mode='random_rate', boost_cost_prob=0.25, observation_mode='partial_5',
                  food_on_death_prob=0.33, food_rate=2.5e-4), "MultiSnake 512 x 25 x 25, 4 snakes, the reference's training dynamics")):
        Ts = 300
        tape = torch.randint(8, (Ts + 10, k_, n_), device=device, dtype=torch.int64)
        ks = [f'agent_{i}' for i in range(k_)]
        ds = [dict(zip(ks, tape[t].unbind(0))) for t in range(Ts + 10)]
        per_call(key, lambda: MultiSnake(n_, k_, s_, device=device, seed=0, **kw_), lambda t: ds[t], d_all, Ts,
                 what + ": `step(a); reset(d['__all__'], return_observations=False)`, one launch per iteration")
        del tape, ds

    # ---- cfg1 (BASELINE configs[0]) per call
    N, T = 64, 2000
    acts = torch.randint(4, (T + 10, N), device=device, dtype=torch.int64)
    per_call('per_call_cfg1_gridworld_64x9',
             lambda: SimpleGridworld(N, 9, start_location=(4, 4), observation_mode='default', device=device, seed=0),
             a1, same, T, 'BASELINE configs[0] (SimpleGridworld 64 x 9 x 9, default observation) on the GPU, '
             '`env.step(a); env.reset(d, return_observations=False)`')

    # ---- the single-agent caller loop as the reference writes it (experiments/main.py:212-227)
    def consistency(env, o):
        env_consistency(env.envs[~o[2].squeeze(-1)])
    for n_envs, T in ((512, 1000), (65536, 60)):
        acts = torch.randint(4, (T + 10, n_envs), device=device, dtype=torch.int64)
        per_call(f'main_py_loop_{n_envs}', lambda: SingleSnake(n_envs, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
                 a1, same, T, 'experiments/main.py:212-227 as written: `step(action); env_consistency(env.envs[~done.squeeze(-1)]); '
                 'reset(done)` (the reset returns its observation; the checker gathers the live envs every step)',
                 reset_kw={}, after_step=consistency)
        per_call(f'main_py_loop_checked_on_device_{n_envs}',
                 lambda: SingleSnake(n_envs, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
                 a1, same, T, 'the same loop with `env.check_consistency(~done)` in place of the gathered copy (the mask of the '
                 'live envs is checked on the device, one `any()` per step)', reset_kw={},
                 after_step=lambda env, o: env.check_consistency(~o[2].squeeze(-1)))

    # ---- cfg3 whole on one GPU and one GPU's share, rollouts; then the per-call loops at cfg3 and cfg2
    rollout('rollout_8192', lambda: SingleSnake(8192, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
            lambda c: (c, 8192), 4, 128, 16, 'BASELINE configs[2] per-GPU share (65536/8), fused rollout, 128 batch-steps per launch',
            traffic_key='rollout_8192x9_chunk128')
    rollout('rollout_cfg3_65536', lambda: SingleSnake(65536, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
            lambda c: (c, 65536), 4, 64, 8, 'BASELINE configs[2] whole (65536 envs) on ONE GPU, fused rollout, 64 batch-steps per launch',
            traffic_key='rollout_65536x9_chunk64')
    for mode, steps in (('default', 16), ('raw', 16), ('positions', 64)):
        rollout(f'rollout_65536x9_gridworld_{mode}',
                lambda: SimpleGridworld(65536, 9, start_location=(4, 4), observation_mode=mode, device=device, seed=0),
                lambda c: (c, 65536), 4, steps, 8, f'SimpleGridworld 65 536 x 9 x 9, observation_mode={mode!r}, fused rollout of '
                f'{steps} batch-steps per launch (one env per lane, gridworld_lane.hip)',
                traffic_key=f'rollout_65536x9_gridworld_{mode}_chunk16')
    rollout('rollout_65536x9_gridworld_default_64steps',
            lambda: SimpleGridworld(65536, 9, start_location=(4, 4), observation_mode='default', device=device, seed=0),
            lambda c: (c, 65536), 4, 64, 4, "SimpleGridworld 65 536 x 9 x 9 'default', fused rollout of 64 batch-steps per launch (4.1 GB of "
            'observations: what a launch costs besides its steps is a quarter of what it is at 16)')
    N, T = 65536, 200
    acts = torch.randint(4, (T + 10, N), device=device, dtype=torch.int64)
    per_call('per_call_gridworld_65536x9_default_no_reset_obs',
             lambda: SimpleGridworld(N, 9, start_location=(4, 4), observation_mode='default', device=device, seed=0),
             a1, same, T, "... `env.step(a); env.reset(d, return_observations=False)`: one observation stream (63.7 MB per call) "
             'instead of two')
    per_call('per_call_gridworld_65536x9_default',
             lambda: SimpleGridworld(N, 9, start_location=(4, 4), observation_mode='default', device=device, seed=0),
             a1, same, T, 'SimpleGridworld 65 536 x 9 x 9 default observation through `env.step(a); env.reset(d)` (one env per '
             'lane from 12 288 envs: gridworld_lane_step_kernel; round 6: on the mirror of one record per env, ONE launch per call; 127 MB of '
             'observations per call with the one reset returns)', reset_kw={}, traffic_key='gridworld_step_65536x9_default_reset_obs')
    per_call('per_call_cfg3_no_mirror', lambda: SingleSnake(N, SIZE, observation_mode=OBS_MODE, device=device, seed=0,
                                                             resident_mirror=False), a1, same, T,
             '`env.step(a); env.reset(d)` at 65 536 envs with the mirror switched off: lane_step_kernel reads the whole (N,3,9,9) '
             'state every call', reset_kw={})
    per_call('per_call_cfg3_no_reset_obs', lambda: SingleSnake(N, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
             a1, same, T, 'BASELINE configs[2] whole through `env.step(a); env.reset(d, return_observations=False)`',
             traffic_key='resident_step_65536x9_partial2')
    per_call('per_call_cfg3', lambda: SingleSnake(N, SIZE, observation_mode=OBS_MODE, device=device, seed=0), a1, same, T,
             "BASELINE configs[2] whole on one GPU through the reference's own call form `env.step(a); env.reset(d)` (resident "
             'mirror, lazy)', reset_kw={}, traffic_key='resident_step_65536x9_partial2_reset_obs')
    for mode in ('default', 'one_channel', 'raw', 'partial_3'):
        per_call(f'per_call_cfg3_{mode}', lambda: SingleSnake(N, SIZE, observation_mode=mode, device=device, seed=0), a1, same, T,
                 f"65 536 x 9 x 9 with observation_mode={mode!r} ('one_channel' is the reference's constructor default) through "
                 '`env.step(a); env.reset(d)` (resident mirror, bit-plane writer)', reset_kw={},
                 traffic_key=f'resident_step_65536x9_{mode}_reset_obs')
    per_call('per_call_cfg3_default_no_reset_obs', lambda: SingleSnake(N, SIZE, observation_mode='default', device=device, seed=0),
             a1, same, T, "65 536 x 9 x 9 'default' through `env.step(a); env.reset(d, return_observations=False)` (round 6: its 16-byte "
             'stores had come out of the compiler as four dword stores each — tools/check_split_stores.py)')
    # ---- the sizes between 9 x 9 and the clock grids on THEIR mirror (round 6: lane_wide_resident.hpp, 48 bytes per env)
    for S_w, mode_w in ((10, 'partial_2'), (11, 'default')):
        per_call(f'per_call_65536x{S_w}_{mode_w}', lambda: SingleSnake(N, S_w, observation_mode=mode_w, device=device, seed=0), a1, same, T,
                 f'65 536 x {S_w} x {S_w} {mode_w} through `env.step(a); env.reset(d)` (resident mirror, lazy)', reset_kw={})
    N, T = 512, 4000
    acts = torch.randint(4, (T + 10, N), device=device, dtype=torch.int64)
    per_call('per_call_512_no_reset_obs', lambda: SingleSnake(N, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
             a1, same, T, 'cfg2 per call: `obs, r, d, info = env.step(a); env.reset(d, return_observations=False)`')
    per_call('per_call_512', lambda: SingleSnake(N, SIZE, observation_mode=OBS_MODE, device=device, seed=0), a1, same, T,
             "cfg2 per call in the reference's own call form: `obs, r, d, info = env.step(a); env.reset(d)`", reset_kw={})
    out['host_calibration_after'] = host_calibration(device)
    # ---- LAST (they leave 20 GB of cached blocks behind, which changes where every later output lands: measured, the cfg4
    # per-call loop went from 34 to 39-41 us behind them): the launch time of the HBM-bound rollouts by output allocation
    alloc_spread('rollout_cfg5_alloc_spread', lambda: SingleSnake(8192, 36, observation_mode='default', device=device, seed=0),
                 lambda c: (c, 8192), 4, 16, 10, 'BASELINE configs[4], the 16-step launch over 10 fresh output allocations (2 GB each)')
    alloc_spread('rollout_cfg3_alloc_spread', lambda: SingleSnake(65536, SIZE, observation_mode=OBS_MODE, device=device, seed=0),
                 lambda c: (c, 65536), 4, 64, 10, 'BASELINE configs[2] whole on one GPU, the 64-step launch over 10 fresh output allocations')
    return out


def key_numbers(line):
    """The numbers the round is judged on, short keys, printed LAST in the line (the driver keeps the tail of stdout)."""
    ex = line.get('extra', {})

    def g(k, f='value'):
        v = ex.get(k, {}).get(f)
        return None if v is None else (float('%.4g' % v) if isinstance(v, float) else v)
    return {
        'cfg2_rollout_eps': float('%.4g' % line['value']),
        'cfg3_strong_scaling_eps': float('%.4g' % line['cfg3_strong_scaling']['value']),
        'cfg3_rollout_eps': g('rollout_cfg3_65536'), 'cfg3_rollout_ms': g('rollout_cfg3_65536', 'ms'),
        'cfg3_rollout_frac_real': g('rollout_cfg3_65536', 'frac_real'),
        'cfg3_share_8192_eps': g('rollout_8192'),
        'cfg3_one_channel_eps': g('rollout_65536x9_one_channel'), 'cfg3_one_channel_frac_real': g('rollout_65536x9_one_channel', 'frac_real'),
        'cfg3_default_eps': g('rollout_65536x9_default'), 'cfg3_default_frac_real': g('rollout_65536x9_default', 'frac_real'),
        'per_call_cfg3_one_channel_us': g('per_call_cfg3_one_channel', 'us'),
        'cfg4_rollout_ms': g('multi_rollout_cfg4_full', 'ms'), 'cfg4_rollout_frac_real': g('multi_rollout_cfg4_full', 'frac_real'),
        'cfg4_rollout_eps': g('multi_rollout_cfg4_full'), 'cfg4_per_call_us': g('per_call_cfg4', 'us'),
        'cfg5_rollout_ms': g('rollout_cfg5_8192x36_default', 'ms'),
        'cfg5_rollout_frac_real': g('rollout_cfg5_8192x36_default', 'frac_real'), 'cfg5_per_call_us': g('per_call_cfg5_8192x36_default', 'us'),
        'gridworld_65536_default_eps': g('rollout_65536x9_gridworld_default'), 'gridworld_65536_default_ms': g('rollout_65536x9_gridworld_default', 'ms'),
        'gridworld_65536_default_frac_real': g('rollout_65536x9_gridworld_default', 'frac_real'),
        'gridworld_65536_per_call_us': g('per_call_gridworld_65536x9_default', 'us'),
        'gridworld_65536_per_call_no_reset_obs_us': g('per_call_gridworld_65536x9_default_no_reset_obs', 'us'),
        'gridworld_65536_default_64steps_ms': g('rollout_65536x9_gridworld_default_64steps', 'ms'),
        'cfg3_raw_eps': g('rollout_65536x9_raw'), 'cfg3_partial3_eps': g('rollout_65536x9_partial_3'),
        'per_call_cfg3_raw_us': g('per_call_cfg3_raw', 'us'), 'per_call_cfg3_partial3_us': g('per_call_cfg3_partial_3', 'us'),
        'cfg1_per_call_us': g('per_call_cfg1_gridworld_64x9', 'us'), 'cfg1_machine': g('per_call_cfg1_gridworld_64x9', 'machine'),
        'per_call_512_us': g('per_call_512', 'us'), 'per_call_512_eps': g('per_call_512'),
        'per_call_512_launches': g('per_call_512', 'launches_per_iter'),
        'per_call_cfg3_us': g('per_call_cfg3', 'us'), 'per_call_cfg3_eps': g('per_call_cfg3'),
        'main_py_loop_512_us': g('main_py_loop_512', 'us'), 'main_py_loop_65536_us': g('main_py_loop_65536', 'us'),
        'a2c_loop_512_eps': g('a2c_loop_512'), 'a2c_fused_actor_512_eps': g('a2c_fused_actor_512'),
        'speeds_rollout_eps': g('multi_rollout_speeds_4096x36_k10'), 'speeds_py_loop_us': g('speeds_py_loop_4096x36_k10', 'us'),
        'cfg4prime_rollout_eps': g('multi_rollout_cfg4prime_partial5'), 'cfg4prime_rollout_ms': g('multi_rollout_cfg4prime_partial5', 'ms'),
        'cfg4prime_per_call_us': g('per_call_cfg4prime_partial5', 'us'),
        'cfg4prime_per_call_split_actions_us': g('per_call_cfg4prime_partial5_split_actions', 'us'),
        'gridworld_65536_per_call_launches': g('per_call_gridworld_65536x9_default', 'launches_per_iter'),
        'multi_512x12_k2_per_call_us': g('per_call_multi_512x12_k2', 'us'),
        'cfg5_rollout_ms_p10_p50_p90': [g('rollout_cfg5_alloc_spread', f) for f in ('ms_p10', 'ms_p50', 'ms_p90')],
        'cfg3_rollout_ms_p10_p50_p90': [g('rollout_cfg3_alloc_spread', f) for f in ('ms_p10', 'ms_p50', 'ms_p90')],
        'multi_512x25_k4_train_per_call_us': g('per_call_multi_512x25_k4_train_partial5', 'us'),
        'per_call_cfg3_default_no_reset_obs_us': g('per_call_cfg3_default_no_reset_obs', 'us'),
        'per_call_s10_partial2_us': g('per_call_65536x10_partial_2', 'us'), 'per_call_s11_default_us': g('per_call_65536x11_default', 'us'),
        's10_partial2_eps': g('rollout_65536x10_partial_2'), 's10_partial2_frac_real': g('rollout_65536x10_partial_2', 'frac_real'),
        's11_default_eps': g('rollout_65536x11_default'), 's11_default_frac_real': g('rollout_65536x11_default', 'frac_real'),
        'box_hbm_fill_TBps': ex.get('host_calibration_after', {}).get('hbm_fill_2GB_TBps'),
    }


def main() -> int:
    args = parse_args()
    in_world = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if args.gpus > 1 and not in_world and not args.worker:
        return spawn_ranks(args)
    return worker(args)


if __name__ == '__main__':
    sys.exit(main())
