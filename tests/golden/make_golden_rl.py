"""Golden vectors for the A2C return computation, recorded from the REAL reference (wurm/rl/a2c.py) in the build
container: inputs (bootstrap, rewards, values, dones, log_probs) and the reference's returns, value loss, policy loss
and — through torch autograd on the reference's own graph — the gradients of (value_loss + policy_loss) with respect
to `values` and `bootstrap_values`.  Data only; see make_golden.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
from wurm.rl import A2C  # noqa: E402


def record(name, T, N, gamma, use_gae, gae_lambda, seed, p_done=0.1, normalise=False):
    g = torch.Generator().manual_seed(seed)
    rewards = (torch.rand((T, N, 1), generator=g) < 0.15).float() - (torch.rand((T, N, 1), generator=g) < 0.05).float()
    values = torch.randn((T, N, 1), generator=g).requires_grad_(True)
    log_probs = -torch.rand((T, N, 1), generator=g) * 2
    dones = torch.rand((T, N, 1), generator=g) < p_done
    bootstrap = torch.randn((N, 1), generator=g).requires_grad_(True)
    a2c = A2C(gamma=gamma, use_gae=use_gae, gae_lambda=gae_lambda, normalise_returns=normalise)
    # `return_returns=True` raises in the reference (tuple += Tensor, a2c.py:76-77): capture the tensor that
    # `returns = torch.stack(returns)` (a2c.py:66) produces by wrapping torch.stack for the duration of the call
    captured = []
    orig_stack = torch.stack

    def stack(tensors, *a, **kw):
        out = orig_stack(tensors, *a, **kw)
        captured.append(out)
        return out

    torch.stack = stack
    try:
        value_loss, policy_loss = a2c.loss(bootstrap, rewards, values, log_probs, dones)
    finally:
        torch.stack = orig_stack
    returns = captured[0]
    if normalise:  # a2c.py:68-69 rebinds `returns`; restate the normalisation on the captured tensor for the record
        returns = (returns - returns.mean()) / (returns.std() + 1e-8)
    (value_loss + policy_loss).backward()
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'), rewards=rewards.numpy(), values=values.detach().numpy(),
        log_probs=log_probs.numpy(), dones=dones.numpy().astype(np.uint8), bootstrap=bootstrap.detach().numpy(),
        returns=returns.detach().numpy(), value_loss=value_loss.detach().numpy(),
        policy_loss=policy_loss.detach().numpy(), grad_values=values.grad.numpy(),
        grad_bootstrap=bootstrap.grad.numpy() if bootstrap.grad is not None else np.zeros((N, 1), np.float32),
        meta=np.array([T, N, int(use_gae), int(normalise)], np.int64),
        gamma=np.array(gamma), gae_lambda=np.array(gae_lambda if gae_lambda is not None else 0.0))
    print(name, 'ok', float(value_loss), float(policy_loss))


SCENARIOS = {
    'a2c_nstep_t40_n64': lambda: record('a2c_nstep_t40_n64', 40, 64, 0.99, False, None, 41),
    'a2c_nstep_t5_n512': lambda: record('a2c_nstep_t5_n512', 5, 512, 0.9, False, None, 42, p_done=0.3),
    'a2c_gae_t40_n64': lambda: record('a2c_gae_t40_n64', 40, 64, 0.99, True, 0.95, 43),
    'a2c_gae_t20_n128': lambda: record('a2c_gae_t20_n128', 20, 128, 0.95, True, 0.8, 44, p_done=0.25),
    'a2c_nstep_norm_t30_n32': lambda: record('a2c_nstep_norm_t30_n32', 30, 32, 0.99, False, None, 45, normalise=True),
}

if __name__ == '__main__':
    for n in (sys.argv[1:] or list(SCENARIOS)):
        SCENARIOS[n]()
