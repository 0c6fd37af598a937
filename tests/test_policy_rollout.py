"""Policy-in-the-loop rollout (include/wurm_hip.h: wurm_single_policy_rollout).

CPU part: the arithmetic spec of oracle/policy.c against torch's fp32 FeedforwardAgent forward (tolerance: 2e-6 absolute
on probabilities, 1e-5 relative on values — the spec fixes the accumulation order, torch's addmm does not), exp_spec
against numpy, and the sampler's distribution.  GPU part: the HIP kernel against the oracle, bit for bit, on every
output; the Python API; the status flag."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.backends import OracleBackend


def _params(E, seed=0, scale=0.3):
    rng = np.random.RandomState(seed)
    return (rng.randn(O.policy_param_count(E)) * scale).astype(np.float32)


def _torch_forward(params, x):
    E = x.shape[1]
    p = torch.tensor(params)
    o = 0

    def take(*shape):
        nonlocal o
        n = int(np.prod(shape))
        t = p[o:o + n].reshape(*shape)
        o += n
        return t
    W1, b1, W2, b2, Wp, bp, Wv, bv = take(64, E), take(64), take(64, 64), take(64), take(4, 64), take(4), take(64), take(1)
    X = torch.tensor(x)
    h = torch.relu(torch.relu(X @ W1.T + b1) @ W2.T + b2)
    return torch.softmax(h @ Wp.T + bp, -1).numpy(), (h @ Wv + bv).numpy()


@pytest.mark.parametrize('E', [3, 27, 75, 147])
def test_spec_forward_matches_torch(E):
    rng = np.random.RandomState(E)
    params = _params(E, seed=E)
    x = rng.choice(np.asarray([0.0, 1.0, 127.0 / 255.0], np.float32), size=(64, E))   # the values an observation takes
    p, v = O.policy_forward(params, x)
    pt, vt = _torch_forward(params, x)
    assert np.abs(p - pt).max() < 2e-6
    assert np.abs(v - vt).max() < 1e-5 * max(1.0, np.abs(vt).max())
    assert np.abs(p.sum(1) - 1).max() < 3e-7


def test_exp_spec():
    x = -np.concatenate([np.linspace(0, 30, 2001), np.logspace(-8, 1.9, 500)]).astype(np.float32)
    got, want = O.exp_spec(x), np.exp(x.astype(np.float64))
    assert np.abs(got / want - 1).max() < 3e-7
    assert O.exp_spec(np.float32(0))[()] == 1.0 and O.exp_spec(np.float32(-100))[()] == 0.0


def test_sampler_distribution():
    """Inverse-CDF sampling with the Philox uniform: chi-square against the probabilities over 40 000 draws."""
    import ctypes
    f = O.lib().oracle_policy_sample
    probs = np.asarray([0.1, 0.2, 0.3, 0.4], np.float32)
    counts = np.zeros(4)
    for env in range(200):
        for call in range(200):
            counts[f(probs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(5), ctypes.c_uint64(2 * call),
                     ctypes.c_uint64(env))] += 1
    expect = probs * counts.sum()
    chi2 = ((counts - expect) ** 2 / expect).sum()
    assert chi2 < 16.3   # 3 dof, p = 0.001


def test_agent_packing_matches_the_spec_layout():
    from wurm_amd.agents import FeedforwardAgent, pack_policy_params
    torch.manual_seed(0)
    agent = FeedforwardAgent(num_actions=4, num_layers=2, hidden_units=64, num_inputs=75)
    params = pack_policy_params(agent).numpy()
    assert params.size == O.policy_param_count(75)
    x = np.random.RandomState(1).rand(16, 75).astype(np.float32)
    p, v = O.policy_forward(params, x)
    with torch.no_grad():
        pt, vt = agent(torch.tensor(x))
    assert np.abs(p - pt.numpy()).max() < 2e-6 and np.abs(v - vt.numpy()[:, 0]).max() < 1e-5
    with pytest.raises(NotImplementedError):
        pack_policy_params(FeedforwardAgent(num_actions=4, num_layers=3, hidden_units=64, num_inputs=75))


# ------------------------------------------------------------------------------------------------------- GPU

@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    x, y = (a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else (a, b)
    assert x.shape == y.shape, f'{what}: shape {x.shape} vs {y.shape}'
    bad = np.argwhere(x != y)
    assert len(bad) == 0, f'{what}: {len(bad)} mismatches, first at {bad[0].tolist()}: {a[tuple(bad[0])]} vs {b[tuple(bad[0])]}'


def _start(backend, N, S, mode):
    envs = np.zeros((N, 3, S, S), np.float32)
    obs = backend.single_reset(envs, np.ones(N, np.uint8), mode)
    return envs, obs


@pytest.mark.gpu
@pytest.mark.parametrize('S,n,T', [(9, 2, 150), (9, 1, 70), (9, 3, 65), (9, 0, 40), (10, 2, 64), (11, 3, 130)])
def test_hip_equals_oracle(hip, S, n, T):
    N, mode = 29, f'partial_{n}'
    E = 3 * (2 * n + 1) ** 2
    params = _params(E, seed=100 + S + n, scale=0.5)
    o, h = OracleBackend(seed=7, env_offset=3), hip(seed=7, env_offset=3)
    envs, obs0 = _start(o, N, S, mode)
    h._next()
    eo, eh = envs.copy(), envs.copy()
    ro, rh = o.single_policy_rollout(eo, obs0, params, T, n), h.single_policy_rollout(eh, obs0, params, T, n)
    assert (rh['status'] == 0).all()
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(eo, eh, 'final state')
    assert ro['done'].sum() > 0 and len(np.unique(ro['actions'])) >= 2
    assert (o.single_check(eo) == 0).all()


@pytest.mark.gpu
def test_status_flags_envs_outside_the_domain(hip):
    N, S, n, T = 12, 9, 2, 20
    params = _params(75, seed=3)
    o, h = OracleBackend(seed=1), hip(seed=1)
    envs, obs0 = _start(o, N, S, 'partial_2')
    h._next()
    envs[4, 0, 2, 2] = 1
    envs[4, 0, 6, 6] = 1              # two foods
    envs[9, 2] = 0                    # no body
    before = envs.copy()
    eh = envs.copy()
    rh = h.single_policy_rollout(eh, obs0, params, T, n)
    assert rh['status'].tolist() == [0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0]
    _same(eh[[4, 9]], before[[4, 9]], 'flagged envs untouched')
    keep = [i for i in range(N) if i not in (4, 9)]
    eo = np.ascontiguousarray(envs[keep])
    # the other envs match an oracle run over just them (env ids are global: compare one by one)
    for idx, i in enumerate(keep):
        oi = OracleBackend(seed=1, env_offset=i)
        oi.call = 1
        ei = envs[i:i + 1].copy()
        ri = oi.single_policy_rollout(ei, obs0[i:i + 1], params, T, n)
        _same(ri['actions'][:, 0], rh['actions'][:, i], f'actions env {i}')
        _same(ri['obs'][:, 0], rh['obs'][:, i], f'obs env {i}')
        _same(ei[0], eh[i], f'state env {i}')


@pytest.mark.gpu
def test_food_on_the_ring_takes_the_generic_loop(hip):
    """Well-formed snakes whose food lies on the border ring are inside the kernel's domain but outside the 9x9 fast
    loop's: same results either way."""
    N, S, n, T = 10, 9, 2, 60
    params = _params(75, seed=8)
    o, h = OracleBackend(seed=2), hip(seed=2)
    envs, _ = _start(o, N, S, 'partial_2')
    h._next()
    for i in (1, 4, 7):
        envs[i, 0] = 0
        envs[i, 0, 0, 3 + i % 3] = 1
    obs0 = _o_observe(envs, 'partial_2')
    eo, eh = envs.copy(), envs.copy()
    ro, rh = o.single_policy_rollout(eo, obs0, params, T, n), h.single_policy_rollout(eh, obs0, params, T, n)
    assert (rh['status'] == 0).all()
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(eo, eh, 'final state')


def _o_observe(envs, mode):
    return O.single_observe(envs, mode)


@pytest.mark.gpu
def test_sharding_invariance(hip):
    """Sampling draws and env randomness are keyed by the global env id: two shards (env_offset 0 / 16) == one batch."""
    N, S, n, T = 32, 9, 2, 80
    params = _params(75, seed=4)
    full = hip(seed=9)
    envs = np.zeros((N, 3, S, S), np.float32)
    obs0 = full.single_reset(envs, np.ones(N, np.uint8), 'partial_2')
    start = envs.copy()
    out = full.single_policy_rollout(envs, obs0, params, T, n)
    for lo in (0, 16):
        shard = hip(seed=9, env_offset=lo)
        shard.call = 1
        es = np.ascontiguousarray(start[lo:lo + 16])
        rs = shard.single_policy_rollout(es, np.ascontiguousarray(obs0[lo:lo + 16]), params, T, n)
        for k in ('actions', 'probs', 'values', 'reward', 'done', 'obs'):
            _same(rs[k], out[k][:, lo:lo + 16], f'shard {lo} {k}')
        _same(es, envs[lo:lo + 16], f'shard {lo} final state')


@pytest.mark.gpu
def test_python_api(hip):
    from wurm_amd.agents import FeedforwardAgent, pack_policy_params
    from wurm_amd.envs import SingleSnake
    torch.manual_seed(3)
    N, T = 40, 33
    env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device='cuda', seed=11)
    agent = FeedforwardAgent(num_actions=4, num_layers=2, hidden_units=64, num_inputs=75).to('cuda')
    state = env.reset()
    start = env.envs.cpu().numpy().copy()
    call0 = env._call
    out = env.policy_rollout(pack_policy_params(agent), state, T)
    assert env._call == call0 + 2 * T
    ref = O.single_policy_rollout(start, state.cpu().numpy().reshape(N, 75), pack_policy_params(agent).cpu().numpy(), T,
                                  obs_n=2, seed=11, call0=call0)
    _same(out['actions'].cpu().numpy(), ref['actions'], 'actions')
    _same(out['probs'].cpu().numpy(), ref['probs'], 'probs')
    _same(out['values'].cpu().numpy(), ref['values'], 'values')
    _same(out['rewards'].cpu().numpy(), ref['reward'], 'rewards')
    _same(out['dones'].cpu().numpy().astype(np.uint8), ref['done'], 'dones')
    _same(out['observations'].cpu().numpy().reshape(T, N, 75), ref['obs'], 'observations')
    _same(env.envs.cpu().numpy(), start, 'final state')
    # the learner's differentiable recomputation agrees with what the kernel acted on
    inputs = torch.cat([state.unsqueeze(0), out['observations'][:-1]]).flatten(2)
    probs, values = agent(inputs)
    assert (probs - out['probs']).abs().max().item() < 2e-6
    assert (values.squeeze(-1) - out['values']).abs().max().item() < 2e-5
    with pytest.raises(NotImplementedError):
        SingleSnake(num_envs=4, size=9, observation_mode='default', device='cuda').policy_rollout(
            pack_policy_params(agent), state[:4], 3)


# ------------------------------------------------------------------------------------------- pinned to the reference

POLICY_FIXTURES = ['policy_ff_n2_s9', 'policy_ff_n1_s10', 'policy_ff_n3_s11', 'policy_ff_n2_s9_sharp']
# Tolerance of the arithmetic spec against the reference's torch forward: the spec fixes the
# accumulation order of every dot product, torch's addmm / softmax do not.
PROB_ATOL, VALUE_RTOL = 2e-6, 1e-5


def _load_policy(name):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.npz'))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize('name', POLICY_FIXTURES)
def test_spec_forward_matches_the_reference_agent(name):
    """oracle/policy.c against probabilities / values recorded from the REAL wurm.agents.FeedforwardAgent
    (tests/golden/make_golden_policy.py; wurm/agents/feedforward.py:8-28) on real observations."""
    fx = _load_policy(name)
    p, v = O.policy_forward(fx['params'], fx['obs'])
    assert np.abs(p - fx['probs']).max() < PROB_ATOL
    assert np.abs(v - fx['values'][:, 0]).max() < VALUE_RTOL * max(1.0, np.abs(fx['values']).max())
    assert (p.argmax(1) == fx['probs'].argmax(1)).all()


@pytest.mark.gpu
@pytest.mark.parametrize('name', POLICY_FIXTURES)
def test_hip_policy_matches_the_reference_agent(name):
    """the fused acting kernel's probs / values of step 0 (policy applied to obs0) against the reference's forward"""
    from tests.hip_backend import HipBackend
    fx = _load_policy(name)
    M, E, n, S = (int(x) for x in fx['meta'])
    h = HipBackend(seed=3)
    envs = np.zeros((M, 3, S, S), np.float32)
    h.single_reset(envs, np.ones(M, np.uint8), 'none')
    out = h.single_policy_rollout(envs, fx['obs'], fx['params'], 1, n)
    assert (out['status'] == 0).all()
    assert np.abs(out['probs'][0] - fx['probs']).max() < PROB_ATOL
    assert np.abs(out['values'][0] - fx['values'][:, 0]).max() < VALUE_RTOL * max(1.0, np.abs(fx['values']).max())
