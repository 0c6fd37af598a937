"""The invariant checkers on FAILING states, pinned to the real reference.

tests/golden/checker_failing_states.npz (tests/golden/make_golden_checker.py) holds valid states corrupted in the ways the
reference's checks look for — two heads, no head, a hole in the body range, the head not at the end, food under the head,
two foods, overlapping snakes, a dead snake that still has cells, ... — and, for each, whether and with WHICH message the
reference raises (wurm/utils.py:113-178, wurm/envs/multi_snake.py:733-769).
  * CPU: the oracle's checker masks (oracle/single_snake.c, oracle/multi_snake.c), turned into a verdict in the reference's
    order of checks, must name the same failure for every state — and the same first failure for the whole batch;
  * GPU: wurm_single_check / wurm_multi_check give the oracle's masks, and the classes' check_consistency() /
    wurm_amd.utils.env_consistency raise (or do not raise) as the reference does, with its message."""
import os

import numpy as np
import pytest

from oracle import oracle as _o

FX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'checker_failing_states.npz'))

# mask bit -> the phrase of the reference's message that identifies the check (messages carry counts: compare phrases)
PHRASES = [(_o.CHK_FOOD_VALUE, 'invalid food pixel'), (_o.CHK_ONE_HEAD, 'multiple num_heads'),
           (_o.CHK_HAS_SNAKE, "don't contain a snake"), (_o.CHK_HEAD_AT_END, 'head not at the end'),
           (_o.CHK_BODY_RANGE, 'inconsistent values'), (_o.CHK_MIN_LENGTH, 'size of less than 3'),
           (_o.CHK_HEAD_ON_FOOD, 'food and head pixel is overlapping'), (_o.CHK_ONE_FOOD, 'exactly one food instance')]


def _verdict(mask: int, one_food: bool) -> str:
    """the phrase of the first failing check in the reference's order ('' = consistent)"""
    for bit, phrase in PHRASES:
        if bit == _o.CHK_ONE_FOOD and not one_food:
            continue
        if mask & bit:
            return phrase
    return ''


def _multi_verdict(mask: int) -> str:
    v = _verdict(mask & 0x7f, one_food=False)
    if v:
        return v
    if mask & 0x100:
        return 'overlapping snakes'
    if mask & 0x200:
        return 'Dead snake contains non-zero elements'
    return ''


def _single_states():
    names = [str(n) for n in FX['single_names']]
    return names, [FX[f'single_env_{i}'] for i in range(len(names))]


def _multi_states():
    names = [str(n) for n in FX['multi_names']]
    K, S = (int(v) for v in FX['multi_shape'])
    out = []
    for i in range(len(names)):
        st = _o.multi_empty_state(1, K, S)
        st['foods'][...] = FX[f'multi_foods_{i}']
        st['heads'][...] = FX[f'multi_heads_{i}']
        st['bodies'][...] = FX[f'multi_bodies_{i}']
        st['dones'][...] = FX[f'multi_dones_{i}']
        out.append(st)
    return names, out


def _agrees(ours: str, reference_message: str, what: str):
    if reference_message == '':
        assert ours == '', f'{what}: the reference accepts this state, the checker says {ours!r}'
    else:
        assert ours != '' and ours in reference_message, f'{what}: reference {reference_message!r}, checker {ours!r}'


def test_fixture_covers_every_check():
    msgs = ' | '.join(str(m) for m in list(FX['single_env_msg']) + list(FX['multi_msg']))
    for _, phrase in PHRASES:
        assert phrase in msgs, phrase
    assert 'overlapping snakes' in msgs and 'Dead snake contains non-zero elements' in msgs
    assert sum(1 for m in FX['single_env_msg'] if str(m) == '') >= 4          # states the reference accepts, too


def test_oracle_single_checker_agrees_with_the_reference_state_by_state():
    names, envs = _single_states()
    for name, e, snake_msg, env_msg in zip(names, envs, FX['single_snake_msg'], FX['single_env_msg']):
        mask = int(_o.single_check(np.ascontiguousarray(e))[0])
        _agrees(_verdict(mask, one_food=False), str(snake_msg), f'snake_consistency {name}')
        _agrees(_verdict(mask, one_food=True), str(env_msg), f'env_consistency {name}')


@pytest.mark.parametrize('S', [9, 12])
def test_oracle_single_checker_on_the_whole_batch(S):
    """the reference checks the batch check by check: its message names the first check ANY env fails"""
    names, envs = _single_states()
    batch = np.ascontiguousarray(np.concatenate([e for n, e in zip(names, envs) if n.startswith(f'S{S}:')]))
    mask = 0
    for m in _o.single_check(batch):
        mask |= int(m)
    _agrees(_verdict(mask, one_food=True), str(FX[f'single_batch_msg_S{S}']), f'batch S={S}')


def test_oracle_multi_checker_agrees_with_the_reference_state_by_state():
    names, states = _multi_states()
    for name, st, msg in zip(names, states, FX['multi_msg']):
        _agrees(_multi_verdict(int(_o.multi_check(st)[0])), str(msg), f'check_consistency {name}')


# ------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_hip_checkers_give_the_oracles_masks_and_the_references_verdicts():
    from tests.hip_backend import HipBackend
    h = HipBackend()
    names, envs = _single_states()
    for name, e, env_msg in zip(names, envs, FX['single_env_msg']):
        mask = int(h.single_check(np.ascontiguousarray(e))[0])
        assert mask == int(_o.single_check(np.ascontiguousarray(e))[0]), name
        _agrees(_verdict(mask, one_food=True), str(env_msg), name)
    mnames, states = _multi_states()
    for name, st, msg in zip(mnames, states, FX['multi_msg']):
        mask = int(h.multi_check(st)[0])
        assert mask == int(_o.multi_check(st)[0]), name
        _agrees(_multi_verdict(mask), str(msg), name)


@pytest.mark.gpu
def test_classes_raise_as_the_reference_does():
    import torch
    from wurm_amd.envs import MultiSnake, SingleSnake
    from wurm_amd.utils import env_consistency, snake_consistency
    dev = torch.device('cuda:0')

    def raised(fn):
        try:
            fn()
            return ''
        except RuntimeError as e:
            return str(e)

    def same(ours, ref, what):
        if ref == '':
            assert ours == '', f'{what}: raised {ours!r}, the reference does not'
        else:  # the reference's text, up to the counts it formats into two of its messages
            key = [p for _, p in PHRASES + [(0, 'overlapping snakes'), (0, 'Dead snake contains non-zero elements')] if p in ref]
            assert key and key[0] in ours, f'{what}: reference {ref!r}, ours {ours!r}'

    names, envs = _single_states()
    for name, e, snake_msg, env_msg in zip(names, envs, FX['single_snake_msg'], FX['single_env_msg']):
        t = torch.tensor(e, device=dev)
        same(raised(lambda: snake_consistency(t)), str(snake_msg), f'snake_consistency {name}')
        same(raised(lambda: env_consistency(t)), str(env_msg), f'env_consistency {name}')
        env = SingleSnake(num_envs=1, size=e.shape[-1], device=dev, manual_setup=True, seed=0)
        env.envs = t.clone()
        same(raised(env.check_consistency), str(env_msg), f'SingleSnake.check_consistency {name}')
    K, S = (int(v) for v in FX['multi_shape'])
    mnames, states = _multi_states()
    for name, st, msg in zip(mnames, states, FX['multi_msg']):
        env = MultiSnake(num_envs=1, num_snakes=K, size=S, device=dev, manual_setup=True, seed=0)
        env.foods, env.heads = torch.tensor(st['foods'], device=dev), torch.tensor(st['heads'], device=dev)
        env.bodies, env.dones = torch.tensor(st['bodies'], device=dev), torch.tensor(st['dones'], device=dev).bool()
        same(raised(env.check_consistency), str(msg), f'MultiSnake.check_consistency {name}')
