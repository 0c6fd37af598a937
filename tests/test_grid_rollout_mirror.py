"""SingleSnake rollouts of grids of 12 x 12 and larger on the mirror of the per-call step (round 6: wurm_single_rollout_resident,
grid_rollout.hip — an env its record describes is read from its clock grid instead of the fp32 planes and written back there,
the planes only while the mirror is not lazy).  Class level: every output of a sequence that mixes rollouts, per-call steps
(with the reset postponed, eager, and missing — a finished env that is stepped again is outside the clock-grid kernels' domain),
looks at the state and in-place edits, with resident_mirror=True / 'eager' against resident_mirror=False (whose kernels the
oracle tests cover: tests/test_hip_vs_oracle.py, test_hip_fused_step.py); and the final launches against the oracle directly.

reference: wurm/envs/single_snake.py:197-304 (step), :322-387 (reset), :130-195 (_observe)."""
import numpy as np
import pytest
import torch

from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu


def _run(policy, N, S, mode, seed):
    from wurm_amd import _lib
    from wurm_amd.envs import SingleSnake
    g = torch.Generator(device='cuda:0').manual_seed(seed)
    tape = torch.randint(0, 4, (10, 7, N), generator=g, device='cuda:0')
    acts = torch.randint(0, 4, (60, N), generator=g, device='cuda:0')
    env = SingleSnake(N, S, observation_mode=mode, device='cuda:0', seed=seed, resident_mirror=policy)
    outs, routes = [], []
    k = 0
    for i in range(10):
        ro = env.rollout(tape[i].clone(), return_observations=(i % 3 != 2))
        routes.append(_lib.lib().wurm_single_last_route().decode())
        outs.append([v.clone() for v in ro.values() if v is not None])
        for j in range(4):
            obs, r, d, info = env.step(acts[k].clone()); k += 1
            outs.append([obs.clone(), r.clone(), d.clone(), info['self_collision'].clone(), info['edge_collision'].clone()])
            if (i + j) % 5 == 4:
                pass                                       # no reset: finished envs are stepped again / rolled out terminal
            elif j % 2:
                back = env.reset(d)
                outs.append([back.clone()])
            else:
                env.reset(d, return_observations=False)
        if i == 3:
            outs.append([env.envs.clone()])                # a look: a lazy mirror is written out
        if i == 5:
            e = env.envs
            e[2, 1] = 0
            e[2, 1, 3, 3] = 1; e[2, 1, 5, 5] = 1           # two heads: outside the domain
            del e
        if i == 7:
            e = env.envs
            e[2] = 0
            e[2, 1, 4, 4] = 1; e[2, 2, 4, 4] = 2; e[2, 2, 4, 3] = 1; e[2, 0, 6, 6] = 1   # a well-formed snake again
            del e
    outs.append([env.envs.clone()])
    return outs, routes, env.mirror_state()


@pytest.mark.parametrize('N,S,mode', [(70, 12, 'partial_2'), (40, 20, 'default'), (24, 36, 'default'), (33, 14, 'raw'),
                                      (20, 36, 'one_channel')])
@pytest.mark.parametrize('policy', [True, 'eager'])
def test_rollouts_and_steps_on_the_mirror_equal_the_planes_only_form(N, S, mode, policy):
    a, ra, ma = _run(policy, N, S, mode, seed=5)
    b, rb, mb = _run(False, N, S, mode, seed=5)
    assert ma['state'] in ('lazy', 'eager') and mb['state'] == 'off'
    assert set(ra) == {'grid_rollout'} and set(rb) == {'grid_rollout'}
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        assert len(x) == len(y)
        for i, (u, v) in enumerate(zip(x, y)):
            assert u.dtype == v.dtype and torch.equal(u, v), f'record {k} output {i}'


@pytest.mark.parametrize('lazy', [True, False])
def test_rollout_resident_against_the_oracle(lazy):
    """the C ABI entry point itself: chained launches on a mirror the first one builds, against the oracle's rollout; the planes
    compared after wurm_single_resident_flush"""
    import ctypes
    from wurm_amd import _lib
    N, S, mode, seed = 50, 16, 'default', 31
    o = OracleBackend(seed=seed, env_offset=3)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    dev = torch.device('cuda:0')
    e_dev = torch.from_numpy(eo.copy()).to(dev)
    l = _lib.lib()
    m, n = _lib.parse_obs_mode(mode)
    nbytes = int(l.wurm_single_resident_size(_lib.i64(N), S, m, n))
    assert nbytes > 0
    res = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    valid = ctypes.c_int(0)
    rng = np.random.RandomState(1)
    call = 40
    stream = _lib.stream_ptr(0)
    for launch in range(6):
        T = int(rng.choice([1, 5, 17, 70]))
        a = rng.randint(0, 4, size=(T, N)).astype(np.int64)
        o.call = call
        ao = a.copy()
        ro = o.single_rollout(eo, ao, mode)
        a_dev = torch.from_numpy(a.copy()).to(dev)
        obs = torch.empty((T, N, 3, S, S), dtype=torch.float32, device=dev)
        reward = torch.empty((T, N), dtype=torch.float32, device=dev)
        fl = torch.empty((3, T, N), dtype=torch.uint8, device=dev)
        rc = l.wurm_single_rollout_resident(_lib.ptr(e_dev), _lib.ptr(a_dev), _lib.ACT_I64, _lib.ptr(reward), _lib.ptr(fl[0]),
                                            _lib.ptr(fl[1]), _lib.ptr(fl[2]), _lib.ptr(obs), m, n, _lib.i64(N), S, _lib.i64(T),
                                            _lib.u64(seed), _lib.u64(call), _lib.i64(3), _lib.ptr(res), ctypes.addressof(valid),
                                            int(lazy), stream)
        assert rc == 0 and valid.value == 1 and l.wurm_single_last_route().decode() == 'grid_rollout'
        call += 2 * T
        assert np.array_equal(ro['obs'].view(np.uint32), obs.cpu().numpy().view(np.uint32)), f'obs launch {launch}'
        assert np.array_equal(ro['reward'], reward.cpu().numpy()) and np.array_equal(ro['done'], fl[0].cpu().numpy())
        assert np.array_equal(ao, a_dev.cpu().numpy()), 'sanitised actions'
    c = _lib.SingleCall()
    c.envs, c.num_envs, c.size = _lib.ptr(e_dev), N, S
    c.resident, c.resident_valid, c.resident_lazy = _lib.ptr(res), valid.value, int(lazy)
    _lib.check(l.wurm_single_resident_flush(ctypes.addressof(c), stream), 'flush')
    assert np.array_equal(eo.view(np.uint32), e_dev.cpu().numpy().view(np.uint32)), 'final state'
