"""The shape-specialised MultiSnake kernels (round 6: K, S and the crop radius compiled in for the shapes the reference's own
experiments run — multi_snake.hip: shape_constants) against the generic kernels, bit for bit: same source, two
instantiations, switched within one process by WURM_MULTI_SHAPE_KERNELS.  The oracle tests at these shapes
(tests/test_hip_multi_vs_oracle.py, test_hip_multi_fused.py, test_multi_resident.py, test_multi_group_rollout.py,
test_full_size_parity.py) run the specialised kernels — the default — so together: specialised == oracle == generic.

reference: wurm/envs/multi_snake.py:462-731 (step), :771-836 (reset), :283-334 (observations);
shapes: experiments/multiagent.py:79-86 + tests/test_multi_snake_env.py:100-104 (4 x 25 x 25, partial_5), BASELINE configs[3]
(4 x 25 x 25 'full'), experiments/speeds.py (10 x 36 x 36)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TRAIN = dict(food_mode='random_rate', respawn_mode='any', boost_cost_prob=0.25, food_on_death_prob=0.33, food_rate=2.5e-3)


def _flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, dict):
        return [t for k in sorted(o) for t in _flat(o[k])]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in _flat(x)]
    return []


def _state(env):
    return [getattr(env, n).clone() for n in ('foods', 'heads', 'bodies', 'dones', 'orientations', 'agent_colours')]


def _routes(kind, N, K, S, T, kw, group_min_envs=None):
    """everything the caller sees over a few launches / iterations, once per setting of the option"""
    from wurm_amd import _lib
    from wurm_amd.envs import MultiSnake
    g = torch.Generator().manual_seed(11)
    tape = torch.randint(8, (3 * T, K, N), generator=g).cuda()
    keys = ['agent_%d' % i for i in range(K)]
    seen = {}
    for v in (0, 1):
        opts = {'WURM_MULTI_SHAPE_KERNELS': v}
        if group_min_envs is not None:
            opts['WURM_MULTI_GROUP_MIN_ENVS'] = group_min_envs
        with _lib.knobs(**opts):
            env = MultiSnake(N, K, S, device='cuda:0', seed=4, **kw)
            outs = []
            if kind == 'rollout':
                for i in range(3):
                    outs += [t.clone() for t in _flat(env.rollout(tape[i * T:(i + 1) * T].contiguous()))]
            else:
                for t in range(3 * T):
                    o = env.step(dict(zip(keys, tape[t])))
                    outs += [x.clone() for x in _flat(o)]
                    r = env.reset(o[2]['__all__'], return_observations=(t % 3 == 0))
                    outs += [x.clone() for x in _flat(r)]
            seen[v] = (outs, _state(env))
    return seen


@pytest.mark.parametrize('kind,N,K,S,T,kw,gmin', [
    ('rollout', 301, 4, 25, 24, dict(observation_mode='partial_5', **TRAIN), None),   # multi_rollout_kernel<.., 4, 25, 5>
    ('percall', 301, 4, 25, 12, dict(observation_mode='partial_5', **TRAIN), None),   # multi_step_kernel<.., PARTIAL, 4, 25, 5>
    ('percall', 301, 4, 25, 12, dict(), None),                                         # multi_step_kernel<.., DEFAULT, 4, 25>
    ('percall', 301, 4, 25, 12, dict(boost=True, **TRAIN), None),
    ('percall', 301, 2, 12, 20, dict(), None),                                         # multi_step_kernel<.., DEFAULT, 2, 12>
    ('percall', 77, 2, 12, 20, dict(**TRAIN), None),
    ('percall', 2051, 4, 25, 6, dict(**TRAIN), None),                                  # ... in the grouped-writer form (>= 2048 envs)
    ('rollout', 83, 4, 25, 10, dict(**TRAIN), 0),                                      # multi_rollout_group_kernel<8,2,1,5,..,4,25>, ragged group
    ('percall', 37, 10, 36, 8, dict(boost=True, **TRAIN), None),                       # multi_step_wg_kernel<.., DEFAULT, 10, 36>
])
def test_specialised_kernels_equal_the_generic_ones(kind, N, K, S, T, kw, gmin):
    seen = _routes(kind, N, K, S, T, kw, gmin)
    a, b = seen[0], seen[1]
    assert len(a[0]) == len(b[0]) and len(a[0]) > 0
    for i, (x, y) in enumerate(zip(a[0], b[0])):
        assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y), (kind, K, S, 'output', i)
    for i, (x, y) in enumerate(zip(a[1], b[1])):
        assert torch.equal(x, y), (kind, K, S, 'state', i)


def test_the_option_is_listed_and_defaults_to_on():
    from wurm_amd import _lib
    assert _lib.lib().wurm_get_option(b'WURM_MULTI_SHAPE_KERNELS') == 1
