"""copy.deepcopy / pickle of env objects (the reference's envs are plain attribute bags that copy and pickle,
wurm/envs/single_snake.py:55-102, multi_snake.py:56-160; here the object also owns ctypes blocks, a step machine, a mirror
and output slabs — derived state that is left out and rebuilt).  On the CPU against the simulating stand-in
(tests/protocol_sim.py): the copy and the original continue identically, whatever was pending when the copy was taken;
on the GPU the same with the real kernels."""
import copy
import pickle

import pytest
import torch

from tests import protocol_enum as pe
from tests.test_protocol_enumeration import _machine_or_skip


@pytest.mark.parametrize('machine', ['python', 'c+torchinfo'])
@pytest.mark.parametrize('mirror', [False, None, 'lazy'])
@pytest.mark.parametrize('kind', ['single', 'grid'])
def test_single_and_gridworld_copy_and_pickle(monkeypatch, kind, mirror, machine):
    _machine_or_skip(machine)
    pe.install_single(monkeypatch, kind, machine)
    for prefix in ((), ('step',), ('step', 'reset_d_noobs'), ('step', 'reset_d', 'step', 'reset_d'), ('step', 'look', 'step')):
        d = pe.make_single(kind, mirror if kind == 'single' else False)
        for ev in prefix:
            getattr(d, ev)()
        for clone in (copy.deepcopy, lambda e: pickle.loads(pickle.dumps(e))):
            d2 = type(d)(clone(d.env), False)
            d2.k = d.k
            assert d2.final() == d.final()                       # state and RNG counter (a postponed reset applied in both)
            a = [d.step(), d.look(), d.reset_none(), d.step()]
            b = [d2.step(), d2.look(), d2.reset_none(), d2.step()]
            assert a == b and d.final() == d2.final(), (prefix, clone)


@pytest.mark.parametrize('machine', ['python', 'c+torchinfo'])
@pytest.mark.parametrize('mirror', [False, None, 'eager'])
def test_multi_snake_copy_and_pickle(monkeypatch, mirror, machine):
    _machine_or_skip(machine)
    pe.install_multi(monkeypatch, True, machine)
    for prefix in ((), ('step',), ('step', 'reset_d_noobs'), ('step', 'reset_d', 'step', 'reset_d'), ('step', 'respawn', 'step')):
        d = pe.make_multi(mirror)
        for ev in prefix:
            getattr(d, ev)()
        for clone in (copy.deepcopy, lambda e: pickle.loads(pickle.dumps(e))):
            d2 = pe.MultiDriver(clone(d.env), False)
            d2.k = d.k
            assert d2.final() == d.final()
            assert d2.env.respawn_mode == d.env.respawn_mode and d2.env.observation_mode == d.env.observation_mode
            a = [d.read_rewards(), d.step(), d.look(), d.reset_none(), d.step(), d.observe()]
            b = [d2.read_rewards(), d2.step(), d2.look(), d2.reset_none(), d2.step(), d2.observe()]
            assert a == b and d.final() == d2.final(), (prefix, clone)


def test_info_of_a_step_pickles_as_a_plain_dict(monkeypatch):
    pe.install_multi(monkeypatch, True, 'python')
    d = pe.make_multi(None)
    a = torch.zeros((pe.M_K, pe.M_N), dtype=torch.long)
    _, _, _, info = d.env.step({'agent_%d' % i: a[i] for i in range(pe.M_K)})
    back = pickle.loads(pickle.dumps(info))
    assert type(back) is dict and list(back) == list(info) and all(torch.equal(back[k], info[k]) for k in info)
    assert len(info) == 5 * pe.M_K and 'size_1' in info and info == dict(info.items())


@pytest.mark.gpu
def test_copies_continue_identically_on_the_gpu():
    from wurm_amd.envs import MultiSnake, SimpleGridworld, SingleSnake
    g = torch.Generator().manual_seed(1)
    env = SingleSnake(300, 9, observation_mode='partial_2', device='cuda:0', seed=3)
    acts = torch.randint(4, (30, 300), generator=g).cuda()
    for t in range(10):
        _, _, d, _ = env.step(acts[t].clone())
        env.reset(d, return_observations=False)              # (postponed: the copy must apply it)
    for clone in (copy.deepcopy, lambda e: pickle.loads(pickle.dumps(e))):
        e2 = clone(env)
        assert torch.equal(e2.envs, env.envs) and e2._call == env._call
        for t in range(10, 20):
            x, y = env.step(acts[t].clone()), e2.step(acts[t].clone())
            assert all(torch.equal(u, v) for u, v in zip(x[:3], y[:3]))
            env.reset(x[2]); e2.reset(y[2])
        assert torch.equal(e2.envs, env.envs)
    grid = SimpleGridworld(64, 9, start_location=(4, 4), device='cuda:0', seed=2)
    g2 = copy.deepcopy(grid)
    a = torch.randint(4, (64,), generator=g).cuda()
    assert all(torch.equal(u, v) for u, v in zip(grid.step(a)[:3], g2.step(a)[:3]))
    K = 3
    menv = MultiSnake(40, K, 12, device='cuda:0', seed=5, food_mode='random_rate', respawn_mode='any', observation_mode='partial_2')
    macts = torch.randint(8, (24, K, 40), generator=g).cuda()
    keys = ['agent_%d' % i for i in range(K)]
    for t in range(8):
        o = menv.step(dict(zip(keys, macts[t])))
        menv.reset(o[2]['__all__'], return_observations=False)
    m2 = copy.deepcopy(menv)
    for t in range(8, 24):
        x, y = menv.step(dict(zip(keys, macts[t]))), m2.step(dict(zip(keys, macts[t])))
        for u, v in zip(x, y):
            assert list(u) == list(v) and all(torch.equal(u[k], v[k]) for k in u)
        menv.reset(x[2]['__all__']); m2.reset(y[2]['__all__'])
    for name in ('foods', 'heads', 'bodies', 'dones', 'orientations', 'agent_colours'):
        assert torch.equal(getattr(menv, name), getattr(m2, name)), name


def test_alias_free_discounts_the_storage_wrapper():
    """torch keeps a storage's Python wrapper alive on the StorageImpl once `untyped_storage()`, `deepcopy`,
    `pickle` or `is_shared()` has made it, and the wrapper owns a reference of the storage — `wurm_torch_alias_free` must not
    take it for a caller's alias (it did: `use_count() == 1`), while a real view still counts."""
    import ctypes
    import gc
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'wurm_amd', 'libwurm_torchinfo.so')
    if not os.path.exists(path):
        pytest.skip('libwurm_torchinfo.so not built')
    fn = ctypes.PyDLL(path).wurm_torch_alias_free
    fn.argtypes, fn.restype = [ctypes.py_object], ctypes.c_int
    makers = [lambda t: None, lambda t: copy.deepcopy(t), lambda t: pickle.dumps(t), lambda t: t.is_shared(),
              lambda t: t.untyped_storage()]
    for make in makers:
        t, u = torch.zeros(12), torch.zeros(3)
        make(t)
        gc.collect()
        assert fn((t, u)) == 1, make
        v = t[2:5]
        assert fn((t, u)) == 0 and fn((u, t)) == 0        # a view is a holder, wrapper or not
        del v
        assert fn((t, u)) == 1
    assert fn([torch.zeros(1)]) == -1 and fn((1,)) == -1


@pytest.mark.gpu
def test_a_copied_env_still_postpones_its_resets():
    """... and with it the whole point: `step; reset(d)` stays ONE launch per iteration on the source and on the
    copy after copy.deepcopy / pickle (both touch the state storages' Python wrappers)."""
    from wurm_amd import _lib
    from wurm_amd.envs import MultiSnake, SingleSnake
    count = _lib.lib().wurm_launch_count
    env = SingleSnake(256, 9, observation_mode='partial_2', device='cuda:0', seed=3)
    acts = torch.randint(4, (40, 256)).cuda()
    K = 2
    menv = MultiSnake(64, K, 12, device='cuda:0', seed=5)
    macts = torch.randint(8, (40, K, 64)).cuda()
    keys = ['agent_%d' % i for i in range(K)]

    def loop_single(e, t0):
        for t in range(t0, t0 + 4):
            d = e.step(acts[t].clone())[2]
            e.reset(d, return_observations=False)
        n0 = count()
        for t in range(t0 + 4, t0 + 10):
            d = e.step(acts[t].clone())[2]
            e.reset(d, return_observations=False)
        return (count() - n0) / 6

    def loop_multi(e, t0):
        for t in range(t0, t0 + 4):
            d = e.step(dict(zip(keys, macts[t])))[2]
            e.reset(d['__all__'], return_observations=False)
        n0 = count()
        for t in range(t0 + 4, t0 + 10):
            d = e.step(dict(zip(keys, macts[t])))[2]
            e.reset(d['__all__'], return_observations=False)
        return (count() - n0) / 6

    assert loop_single(env, 0) == 1.0 and loop_multi(menv, 0) == 1.0
    for clone in (copy.deepcopy, lambda e: pickle.loads(pickle.dumps(e))):
        e2, m2 = clone(env), clone(menv)
        assert loop_single(env, 10) == 1.0 and loop_single(e2, 10) == 1.0, clone
        assert loop_multi(menv, 10) == 1.0 and loop_multi(m2, 10) == 1.0, clone
