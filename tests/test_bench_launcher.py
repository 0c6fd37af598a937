"""bench.py --gpus N must start its own N ranks: the parent spawns a fresh torch.distributed.run
child before touching any GPU and relays rank 0's JSON line.  Exercised here with --dry-run (gloo on CPU, the rollout
launch replaced by a sleep): rendezvous, barrier, max-over-ranks time, summed env-steps, one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags], capture_output=True, text=True,
                       env=e, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    d = _run('--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1')
    assert d['n_gpus'] == 2 and d['config']['world_size'] == 2 and d['config']['backend'] == 'gloo'
    assert d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
    # whole-job value: both ranks' env-steps over the max-over-ranks time
    per_launch = d['config']['env_steps_per_launch_per_gpu']
    assert abs(d['value'] - 2 * 3 * per_launch / (d['ms_per_step'] * 1e-3 * 3)) / d['value'] < 1e-6
    assert 'cpu_baseline' not in d and 'extra' not in d
    # BASELINE configs[2] split over the two ranks rides along with the weak-scaling headline
    c3 = d['cfg3_strong_scaling']
    assert c3['global_num_envs'] == 65536 and c3['num_envs_per_gpu'] == 32768 and c3['scaling'] == 'strong'
    assert c3['value'] > 0 and c3['steps'] == 3


def test_cfg3_is_split_over_the_ranks():
    d = _run('--gpus', '2', '--dry-run', '--steps', '2', '--warmup', '0', '--workload', 'cfg3')
    assert d['config']['global_num_envs'] == 65536 and d['config']['num_envs_per_gpu'] == 32768
    assert d['scaling'] == 'strong' and d['config']['baseline_config'] == 'BASELINE.json configs[2]'
    assert abs(d['cfg3_strong_scaling']['value'] - d['value']) / d['value'] < 1e-9   # the same region, not timed twice


def test_single_process_default_and_cpu_baseline_legs():
    d = _run('--dry-run', '--steps', '2', '--warmup', '1')
    assert d['n_gpus'] == 1 and d['config']['world_size'] == 1
    assert str(d['config']['batch_steps_per_launch']) in d['config']['workload']
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['cores'] >= 1
    assert cb['c_port_1thread']['value'] > 0 and cb['c_port_allcores']['value'] > 0


def test_worker_under_an_external_torchrun():
    """the driver's own launch line for N > 1: python -m torch.distributed.run ... bench.py --gpus N"""
    from bench import free_port
    e = dict(os.environ)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
                        os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    assert json.loads(lines[0])['n_gpus'] == 2


def test_world_size_8_with_an_uneven_global_batch():
    """the 8-GPU shape north_star names, on gloo: 65 537 envs do not divide by 8 — every rank gets a contiguous block,
    the blocks tile the batch, the line's key numbers carry the strong-scaling value"""
    d = _run('--gpus', '8', '--dry-run', '--steps', '2', '--warmup', '0', '--workload', 'cfg3', '--num-envs', '65537')
    assert d['n_gpus'] == 8 and d['config']['world_size'] == 8 and d['scaling'] == 'strong'
    assert d['config']['global_num_envs'] == 65537
    assert d['config']['num_envs_per_gpu'] == 8193            # rank 0's block: the remainder goes to the first ranks
    assert d['value'] > 0 and d['key']['cfg3_strong_scaling_eps'] > 0
    from wurm_amd.sharding import shard_range
    blocks = [shard_range(65537, r, 8) for r in range(8)]
    assert blocks[0][0] == 0 and sum(n for _, n in blocks) == 65537
    assert all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(7))
    assert max(n for _, n in blocks) - min(n for _, n in blocks) == 1
    # every rank's own record rides in the line (device, shard, communicator init, launch time): VERDICT r04 #9
    ranks = d['ranks']
    assert [r['rank'] for r in ranks] == list(range(8))
    assert [(r['env_offset'], r['num_envs']) for r in ranks] == blocks
    assert all(r['comm_init_s'] >= 0 and r['avg_launch_ms'] > 0 for r in ranks)
