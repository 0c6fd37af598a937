"""The N>1 path on CPU: two processes over gloo exercise wurm_amd.sharding (block partition, shared seed, stats
all-reduce, output gather) and prove the sharding scheme itself — each rank steps ITS block of global env ids with
`env_offset` (here through the CPU oracle, which implements the same RNG keying as the kernels) and the gathered
trajectories equal the unsharded run bit-for-bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from wurm_amd.sharding import shard_range
    for n in (0, 1, 7, 512, 65536, 65537):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (o0, c0), (o1, _) in zip(blocks, blocks[1:]):
                assert o0 + c0 == o1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world_size, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    try:
        from wurm_amd import sharding
        from oracle import oracle

        assert sharding.world() == (rank, world_size)
        torch.manual_seed(100 + rank)  # ranks disagree locally ...
        seed = sharding.shared_seed(None)
        seeds = [None] * world_size
        dist.all_gather_object(seeds, seed)
        assert len(set(seeds)) == 1  # ... but agree after the broadcast

        N, S, T = 37, 9, 60  # uneven split: 19 + 18
        offset, count = sharding.shard_range(N, rank, world_size)
        rng = np.random.RandomState(0)
        actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
        envs = np.zeros((count, 3, S, S), np.float32)
        oracle.single_reset(envs, np.ones(count, np.uint8), 'none', seed=seed, call=0, env_offset=offset)
        mine = np.ascontiguousarray(actions[:, offset:offset + count])
        out = oracle.single_rollout(envs, mine, 'partial_2', seed=seed, call0=1, env_offset=offset)

        stats = sharding.RolloutStats()
        stats.add(torch.from_numpy(out['reward']), torch.from_numpy(out['done']),
                  torch.from_numpy(out['self_collision']), torch.from_numpy(out['edge_collision']))
        total = stats.all_reduce()
        assert total['env_steps'] == N * T

        g_state = sharding.gather_env_dim(torch.from_numpy(envs), N, dim=0)
        g_obs = sharding.gather_env_dim(torch.from_numpy(out['obs']), N, dim=1)
        g_done = sharding.gather_env_dim(torch.from_numpy(out['done']), N, dim=1)
        g_rew = sharding.gather_env_dim(torch.from_numpy(out['reward']), N, dim=1)
        if rank == 0:
            ref = np.zeros((N, 3, S, S), np.float32)
            oracle.single_reset(ref, np.ones(N, np.uint8), 'none', seed=seed, call=0)
            a = actions.copy()
            exp = oracle.single_rollout(ref, a, 'partial_2', seed=seed, call0=1)
            assert np.array_equal(g_state.numpy(), ref)
            assert np.array_equal(g_obs.numpy().view(np.uint32), exp['obs'].view(np.uint32))
            assert np.array_equal(g_done.numpy(), exp['done'])
            assert total['episodes'] == float(exp['done'].sum())
            assert total['reward_sum'] == float(exp['reward'].sum())
            assert np.array_equal(g_rew.numpy(), exp['reward'])
            open(os.path.join(tmp, 'ok'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_two_rank_batch_split(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok').exists()


def test_make_sharded_passes_offset_and_seed():
    from wurm_amd import sharding

    class Probe(object):
        def __init__(self, num_envs, size, seed=None, env_offset=0, device='cpu'):
            self.num_envs, self.size, self.seed, self.env_offset = num_envs, size, seed, env_offset

    shards = [sharding.make_sharded(Probe, 65536, 9, seed=5, rank=r, world_size=8) for r in range(8)]
    assert [s.num_envs for s in shards] == [8192] * 8
    assert [s.env_offset for s in shards] == [8192 * r for r in range(8)]
    assert {s.seed for s in shards} == {5}
