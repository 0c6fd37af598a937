"""Batch-split of environments over the GPUs of one node.

Environments never interact (the reference's only cross-env couplings — the shared torch RNG stream, wurm/utils.py:224,
and the `torch.any(boosted_agents)` gate, wurm/envs/multi_snake.py:503 — have no per-env semantic effect), so the
path shards with NO data-path collective: rank r owns the contiguous block of global env ids
[offset_r, offset_r + count_r) and passes `env_offset=offset_r`; the kernels key every random draw by the global id,
so trajectories are identical for any number of ranks.  RCCL (torch.distributed, backend 'nccl' on MI355X; 'gloo' in
the CPU tests) is used only OUTSIDE the step loop: to agree on the seed, to sum episode statistics, and optionally
to gather per-step outputs for a single learner.
"""
from typing import Optional, Tuple

import torch


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def world() -> Tuple[int, int]:
    """(rank, world_size) of the default process group, (0, 1) when not distributed."""
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def shard_range(global_num_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """(offset, count) of rank's contiguous block; the first `global_num_envs % world_size` ranks get one more."""
    if not 0 <= rank < world_size:
        raise ValueError(f'rank {rank} outside world of {world_size}')
    base, rem = divmod(global_num_envs, world_size)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def shared_seed(seed: Optional[int] = None, device: str = 'cpu') -> int:
    """One seed for all ranks: rank 0's value (drawn from torch's generator if None) is broadcast."""
    d = _dist()
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    if d is None:
        return int(seed)
    t = torch.tensor([seed], dtype=torch.int64, device=device)
    d.broadcast(t, src=0)
    return int(t.item())


def make_sharded(env_cls, global_num_envs: int, *args, seed: Optional[int] = None, rank: Optional[int] = None,
                 world_size: Optional[int] = None, **kwargs):
    """This rank's shard of a batch of `global_num_envs` envs: env_cls(num_envs=count, ..., seed, env_offset)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    offset, count = shard_range(global_num_envs, rank, world_size)
    if _dist() and _dist().get_backend() == 'nccl':
        # NCCL / RCCL broadcasts device tensors only; the env classes default to the current HIP device
        bdev = torch.device(kwargs.get('device', 'cuda'))
        if bdev.type != 'cuda':
            raise ValueError('make_sharded under the nccl backend needs a HIP device')
        if bdev.index is None:
            bdev = torch.device('cuda', torch.cuda.current_device())
    else:
        bdev = 'cpu'
    return env_cls(count, *args, seed=shared_seed(seed, device=bdev), env_offset=offset, **kwargs)


class RolloutStats(object):
    """Node-level counters of a rollout chunk, summed over ranks with one small all-reduce per chunk."""

    FIELDS = ('env_steps', 'episodes', 'reward_sum', 'self_collisions', 'edge_collisions')

    def __init__(self, device='cpu'):
        self.device = device
        self.values = torch.zeros(len(self.FIELDS), dtype=torch.float64, device=device)

    def add(self, rewards: torch.Tensor, dones: torch.Tensor, self_collision: Optional[torch.Tensor] = None,
            edge_collision: Optional[torch.Tensor] = None):
        """rewards / dones of any shape (..., N): every element is one env-step."""
        v = [float(dones.numel()), dones.sum(), rewards.sum(),
             self_collision.sum() if self_collision is not None else 0.0,
             edge_collision.sum() if edge_collision is not None else 0.0]
        self.values += torch.stack([torch.as_tensor(x, dtype=torch.float64, device=self.device) for x in v])

    def all_reduce(self) -> dict:
        t = self.values.clone()
        d = _dist()
        if d is not None:
            d.all_reduce(t)  # SUM
        return dict(zip(self.FIELDS, t.tolist()))


def gather_env_dim(t: torch.Tensor, global_num_envs: int, dim: int = 0) -> torch.Tensor:
    """All-gathers a per-shard tensor along its env dimension into the global batch order (for a single learner).
    Shards may be uneven (shard_range)."""
    d = _dist()
    if d is None:
        return t
    rank, w = d.get_rank(), d.get_world_size()
    t = t.movedim(dim, 0).contiguous()
    counts = [shard_range(global_num_envs, r, w)[1] for r in range(w)]
    if len(set(counts)) == 1:
        out = torch.empty((global_num_envs,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        d.all_gather_into_tensor(out, t)
    else:  # collectives need equal sizes: pad every shard to the largest block, gather, drop the padding
        m = max(counts)
        padded = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        padded[:counts[rank]] = t
        buf = torch.empty((w * m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        d.all_gather_into_tensor(buf, padded)
        out = torch.cat([buf[r * m:r * m + counts[r]] for r in range(w)], dim=0)
    return out.movedim(0, dim)
