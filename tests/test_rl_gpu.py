"""GPU parity of the learner-side glue: A2C return scan (forward bit-exact vs oracle and vs the
reference's recorded returns; backward vs the gradients torch autograd produced on the reference's own graph),
the A2C loss values, the preallocated TrajectoryStore and the fused logging statistics."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import replay

pytestmark = pytest.mark.gpu
DEV = 'cuda'
RL = ['a2c_nstep_t40_n64', 'a2c_nstep_t5_n512', 'a2c_gae_t40_n64', 'a2c_gae_t20_n128', 'a2c_nstep_norm_t30_n32']


@pytest.mark.parametrize('name', RL)
def test_a2c_loss_matches_reference(name):
    from wurm_amd.rl import A2C
    z = replay.load(name)
    T, N, gae, norm = (int(v) for v in z['meta'])
    rewards = torch.from_numpy(z['rewards']).to(DEV)
    values = torch.from_numpy(z['values']).to(DEV).requires_grad_(True)
    log_probs = torch.from_numpy(z['log_probs']).to(DEV)
    dones = torch.from_numpy(z['dones']).to(DEV).bool()
    bootstrap = torch.from_numpy(z['bootstrap']).to(DEV).requires_grad_(True)
    a2c = A2C(gamma=float(z['gamma']), use_gae=bool(gae), gae_lambda=float(z['gae_lambda']) if gae else None,
              normalise_returns=bool(norm))
    value_loss, policy_loss, returns = a2c.loss(bootstrap, rewards, values, log_probs, dones, return_returns=True)
    got = returns.detach().cpu().numpy()
    if norm:
        assert np.allclose(got, z['returns'], rtol=0, atol=2e-6)
    else:  # fp32 scan in the reference's operation order: bit-exact
        assert np.array_equal(got.view(np.uint32), z['returns'].view(np.uint32))
    # the two mean reductions are torch ops on the GPU: summation order differs from torch-CPU -> tolerance 1e-6 rel
    assert abs(value_loss.item() - float(z['value_loss'])) <= 2e-6 * max(1.0, abs(float(z['value_loss'])))
    assert abs(policy_loss.item() - float(z['policy_loss'])) <= 2e-6 * max(1.0, abs(float(z['policy_loss'])))
    (value_loss + policy_loss).backward()
    assert np.allclose(values.grad.cpu().numpy(), z['grad_values'], rtol=1e-4, atol=1e-7)
    gb = bootstrap.grad.cpu().numpy() if bootstrap.grad is not None else np.zeros_like(z['grad_bootstrap'])
    assert np.allclose(gb, z['grad_bootstrap'], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('T,N,gae', [(1, 1, False), (64, 8192, False), (64, 8192, True), (7, 513, True)])
def test_a2c_returns_vs_oracle(T, N, gae):
    from wurm_amd.rl import a2c_returns
    rng = np.random.RandomState(T * 31 + N)
    rewards = rng.randn(T, N).astype(np.float32)
    values = rng.randn(T, N).astype(np.float32)
    dones = rng.rand(T, N) < 0.1
    boot = rng.randn(N).astype(np.float32)
    want = oracle.a2c_returns(boot, rewards, values, dones, 0.99, gae, 0.95 if gae else None)
    got = a2c_returns(torch.from_numpy(boot).to(DEV), torch.from_numpy(rewards).to(DEV),
                      torch.from_numpy(values).to(DEV), torch.from_numpy(dones).to(DEV), 0.99, gae,
                      0.95 if gae else None).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_trajectory_store():
    from wurm_amd.rl import TrajectoryStore
    store = TrajectoryStore(capacity=4)
    w = torch.ones(3, 1, device=DEV, requires_grad=True)
    rows = []
    for t in range(6):  # beyond the initial capacity: grows
        v = w * (t + 1)
        rows.append(v)
        store.append(value=v, reward=torch.full((3, 1), float(t), device=DEV), done=torch.zeros(3, 1, dtype=torch.bool, device=DEV))
    assert store.values.shape == (6, 3, 1) and store.rewards.shape == (6, 3, 1) and store.dones.dtype == torch.bool
    assert torch.equal(store.values, torch.stack(rows))
    store.values.sum().backward()  # autograd flows through the buffer copies
    assert torch.equal(w.grad, torch.full((3, 1), 21.0, device=DEV))
    store.clear()
    with pytest.raises(RuntimeError):
        _ = store.values
    store.append(value=(w * 2).detach())
    assert store.values.shape == (1, 3, 1)


def test_single_stats():
    from wurm_amd import _lib
    from wurm_amd.envs import SingleSnake
    env = SingleSnake(num_envs=300, size=12, observation_mode='default', seed=4)
    acc = torch.zeros(5, dtype=torch.float64, device=DEV)
    want = np.zeros(5)
    actions = torch.randint(4, (20, 300), device=DEV)
    for t in range(20):
        obs, reward, done, info = env.step(actions[t])
        rc = _lib.lib().wurm_single_stats(_lib.ptr(env.envs), _lib.ptr(reward), _lib.ptr(done), _lib.ptr(info['self_collision']),
                                          _lib.ptr(info['edge_collision']), _lib.ptr(acc), _lib.i64(300), 12, _lib.stream_ptr())
        assert rc == 0
        want += oracle.single_stats(env.envs.cpu().numpy(), reward.cpu().numpy().reshape(-1), done.cpu().numpy().reshape(-1),
                                    info['self_collision'].cpu().numpy(), info['edge_collision'].cpu().numpy())
        env.reset(done)
    assert np.array_equal(acc.cpu().numpy(), want)
