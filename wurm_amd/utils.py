"""Tensor helpers of the env path — same names and meaning as the reference's wurm/utils.py:24-178.

The invariant checkers run as one fused reduction kernel per call (wurm_single_check in include/wurm_hip.h)
that returns a per-env error bitmask; the exception types and messages are the reference's.
"""
import torch

from wurm_amd import _lib
from wurm_amd.constants import FOOD_CHANNEL, HEAD_CHANNEL, BODY_CHANNEL

CHK_FOOD_VALUE, CHK_ONE_HEAD, CHK_HAS_SNAKE, CHK_HEAD_AT_END, CHK_BODY_RANGE, CHK_MIN_LENGTH, CHK_HEAD_ON_FOOD, \
    CHK_ONE_FOOD = (1 << i for i in range(8))


def food(envs: torch.Tensor) -> torch.Tensor:
    """reference wurm/utils.py:24-25"""
    return envs[:, FOOD_CHANNEL:FOOD_CHANNEL + 1]


def head(envs: torch.Tensor) -> torch.Tensor:
    """reference wurm/utils.py:28-29"""
    return envs[:, HEAD_CHANNEL:HEAD_CHANNEL + 1]


def body(envs: torch.Tensor) -> torch.Tensor:
    """reference wurm/utils.py:32-33"""
    return envs[:, BODY_CHANNEL:BODY_CHANNEL + 1]


def get_test_env(size: int, orientation: str = 'up') -> torch.Tensor:
    """Predetermined single-snake boards used by the known-answer tests; the same boards as the reference's
    fixture function (wurm/utils.py:68-110), written as data: (tail..head cells, food cell)."""
    boards = {
        'up': ([(3, 3), (3, 4), (4, 4), (5, 4)], (6, 6)),
        'right': ([(3, 3), (3, 4), (4, 4), (4, 5)], (6, 9)),
        'down': ([(8, 8), (7, 8), (6, 8), (5, 8)], (7, 2)),
        'left': ([(8, 7), (7, 7), (6, 7), (6, 6)], (1, 2)),
    }
    if orientation not in boards:
        raise Exception
    cells, food_cell = boards[orientation]
    env = torch.zeros((1, 3, size, size))
    for i, (y, x) in enumerate(cells):
        env[0, BODY_CHANNEL, y, x] = i + 1
    env[0, HEAD_CHANNEL, cells[-1][0], cells[-1][1]] = 1
    env[0, FOOD_CHANNEL, food_cell[0], food_cell[1]] = 1
    return env


def consistency_mask(envs: torch.Tensor) -> torch.Tensor:
    """Per-env uint32 error bitmask (CHK_* bits) of a (n,3,S,S) single-snake state tensor on the GPU."""
    if envs.dim() != 4 or envs.shape[1] != 3 or envs.shape[2] != envs.shape[3]:
        raise RuntimeError('expected a (n, 3, size, size) tensor')
    if envs.device.type != 'cuda':
        raise _lib.WurmHipError('consistency checks run on the GPU: envs must be a device tensor')
    e = envs.to(torch.float32).contiguous()
    n, S = e.shape[0], e.shape[2]
    err = torch.empty(n, dtype=torch.int32, device=e.device)
    rc = _lib.call(e.device.index, _lib.lib().wurm_single_check, _lib.ptr(e), _lib.ptr(err), _lib.i64(n), S, _lib.stream_ptr(e.device.index))
    _lib.check(rc, 'env_consistency')
    return err


def _raise_for(mask: int, one_food: bool):
    # same order and messages as the reference (wurm/utils.py:119-178)
    if mask & CHK_FOOD_VALUE:
        raise RuntimeError('An environment has an invalid food pixel')
    if mask & CHK_ONE_HEAD:
        raise RuntimeError('An environment has multiple num_heads for a single snake.')
    if mask & CHK_HAS_SNAKE:
        raise RuntimeError('Some environments don\'t contain a snake.')
    if mask & CHK_HEAD_AT_END:
        raise RuntimeError('An environment has a snake with it\'s head not at the end of the body.')
    if mask & CHK_BODY_RANGE:
        raise RuntimeError('An environment has a body with inconsistent values i.e. not range(n)')
    if mask & CHK_MIN_LENGTH:
        raise RuntimeError('A snake has size of less than 3.')
    if mask & CHK_HEAD_ON_FOOD:
        raise RuntimeError('A food and head pixel is overlapping in some env(s).')
    if one_food and (mask & CHK_ONE_FOOD):
        raise RuntimeError('An environment doesn\'t contain exactly one food instance')


def snake_consistency(envs: torch.Tensor):
    """Checks for consistency of a 3 channel single-snake env (reference wurm/utils.py:113-164)."""
    if envs.shape[0] == 0:
        return
    _raise_for(_or_reduce(consistency_mask(envs)), one_food=False)


def env_consistency(envs: torch.Tensor):
    """Runs multiple checks for environment consistency and throws an exception if any fail
    (reference wurm/utils.py:167-178)."""
    if envs.shape[0] == 0:
        return
    _raise_for(_or_reduce(consistency_mask(envs)), one_food=True)


def _or_reduce(mask: torch.Tensor, nbits: int = 8) -> int:
    # OR over envs of a bit mask.  The usual answer is 0: one `any()` (one small kernel, one sync — the reference's
    # checks sync several times) settles that; only a failing batch pays for the per-bit reduction
    if not bool(mask.any()):
        return 0
    bits = (mask.unsqueeze(-1) >> torch.arange(nbits, device=mask.device, dtype=torch.int32)) & 1
    present = bits.any(dim=0).cpu().tolist()
    return sum((1 << i) for i, p in enumerate(present) if p)


def determine_orientations(envs: torch.Tensor) -> torch.Tensor:
    """Returns a batch of snake orientations {0,1,2,3} from a (n,3,size,size) batch of envs
    (reference wurm/utils.py:36-65): orientation o <=> the head sits at neck + [(-1,0),(0,+1),(+1,0),(0,-1)][o]."""
    if envs.dim() != 4 or envs.shape[1] != 3 or envs.shape[2] != envs.shape[3]:
        raise RuntimeError('expected a (n, 3, size, size) tensor')
    if envs.device.type != 'cuda':
        raise _lib.WurmHipError('determine_orientations runs on the GPU: envs must be a device tensor')
    e = envs.to(torch.float32).contiguous()
    n, S = e.shape[0], e.shape[2]
    out = torch.empty(n, dtype=torch.long, device=e.device)
    rc = _lib.call(e.device.index, _lib.lib().wurm_orientations, _lib.ptr(e), _lib.ptr(out), _lib.i64(n), S, _lib.stream_ptr(e.device.index))
    _lib.check(rc, 'determine_orientations')
    return out
