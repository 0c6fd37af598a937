"""A2C loss with the return computation on the GPU — same class interface as the reference's wurm.rl.A2C
(wurm/rl/a2c.py:9-79).

The reverse scan over time that builds `returns` (n-step discounted returns :60-64, GAE :50-59) runs as one HIP kernel
(`wurm_a2c_returns`, bit-identical fp32 to the reference's op sequence) instead of a Python loop of 3-6 torch ops per
time step; a second kernel provides its gradient, so the op sits in the autograd graph exactly where the reference's
torch ops do (the reference does not detach `returns`: with GAE, and through `bootstrap_values`, gradients flow
through it into the value head).  The two reductions that finish the loss (:71-73) stay torch ops.
"""
import ctypes
from typing import Callable

import torch
import torch.nn.functional as F

from wurm_amd import _lib

EPS = 1e-8


class _Returns(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bootstrap, rewards, values, dones, gamma, use_gae, gamma_lambda):
        T, N = rewards.shape
        returns = torch.empty((T, N), dtype=torch.float32, device=rewards.device)
        rc = _lib.call(rewards.device.index, _lib.lib().wurm_a2c_returns, _lib.ptr(bootstrap), _lib.ptr(rewards), _lib.ptr(values), _lib.ptr(dones),
                                         ctypes.c_float(gamma), int(use_gae), ctypes.c_float(gamma_lambda),
                                         _lib.ptr(returns), _lib.i64(T), _lib.i64(N),
                                         _lib.stream_ptr(rewards.device.index))
        _lib.check(rc, 'A2C returns')
        ctx.save_for_backward(dones)
        ctx.args = (gamma, use_gae, gamma_lambda)
        return returns

    @staticmethod
    def backward(ctx, grad_returns):
        (dones,) = ctx.saved_tensors
        gamma, use_gae, gamma_lambda = ctx.args
        T, N = dones.shape
        g = grad_returns.contiguous().to(torch.float32)
        need_b, _, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        grad_values = torch.empty((T, N), dtype=torch.float32, device=g.device) if need_v else None
        grad_boot = torch.empty(N, dtype=torch.float32, device=g.device) if need_b else None
        rc = _lib.call(g.device.index, _lib.lib().wurm_a2c_returns_backward, _lib.ptr(g), _lib.ptr(dones), ctypes.c_float(gamma), int(use_gae),
                                                  ctypes.c_float(gamma_lambda), _lib.ptr(grad_values),
                                                  _lib.ptr(grad_boot), _lib.i64(T), _lib.i64(N),
                                                  _lib.stream_ptr(g.device.index))
        _lib.check(rc, 'A2C returns backward')
        return grad_boot, None, grad_values, None, None, None, None


def a2c_returns(bootstrap_values: torch.Tensor, rewards: torch.Tensor, values: torch.Tensor, dones: torch.Tensor,
                gamma: float, use_gae: bool = False, gae_lambda: float = None) -> torch.Tensor:
    """`returns` of wurm/rl/a2c.py:49-66 with the shape of `rewards` ((T,N) or (T,N,1)); differentiable w.r.t. `values`
    (GAE) and `bootstrap_values`."""
    if rewards.device.type != 'cuda':
        raise _lib.WurmHipError('a2c_returns runs on the GPU: tensors must be device tensors')
    shape = rewards.shape
    T = shape[0]
    N = rewards.numel() // max(T, 1)
    r = rewards.reshape(T, N).to(torch.float32).contiguous()
    v = values.reshape(T, N).to(torch.float32).contiguous()
    b = bootstrap_values.reshape(N).to(torch.float32).contiguous()
    d = dones.reshape(T, N)
    d = (d if d.dtype == torch.bool else d != 0).contiguous()
    import numpy as np
    g = float(np.float32(gamma))
    gl = float(np.float32(gamma * gae_lambda)) if use_gae else 0.0  # :56 gamma * lambda is one python float
    return _Returns.apply(b, r, v, d, g, bool(use_gae), gl).reshape(shape)


class A2C(object):
    """Class that encapsulates the advantage actor-critic algorithm (reference wurm/rl/a2c.py:9-30).

    Args:
        gamma: Discount value
        value_loss_fn: Loss function between values and returns i.e. Huber, MSE
        normalise_returns: Whether or not to normalise target returns
    """
    def __init__(self,
                 gamma: float,
                 value_loss_fn: Callable = F.smooth_l1_loss,
                 normalise_returns: bool = False,
                 use_gae: bool = False,
                 gae_lambda: float = None,
                 dtype: torch.dtype = torch.float):
        self.gamma = gamma
        self.normalise_returns = normalise_returns
        self.use_gae = use_gae
        self.gae_lambda = gae_lambda
        self.value_loss_fn = value_loss_fn
        self.dtype = dtype

    def loss(self,
             bootstrap_values: torch.Tensor,
             rewards: torch.Tensor,
             values: torch.Tensor,
             log_probs: torch.Tensor,
             dones: torch.Tensor,
             return_returns: bool = False):
        """Calculate A2C loss (reference :32-79).  Tensors are (num_steps, num_envs[, 1]); bootstrap_values
        (num_envs[, 1]).  `return_returns=True` appends the returns tensor (the reference's own `ret += returns` raises
        a TypeError at :77)."""
        returns = a2c_returns(bootstrap_values, rewards, values, dones, self.gamma, self.use_gae, self.gae_lambda)

        if self.normalise_returns:
            returns = (returns - returns.mean()) / (returns.std() + EPS)

        value_loss = self.value_loss_fn(values, returns).mean()
        advantages = returns - values
        policy_loss = - (advantages.detach() * log_probs).mean()

        ret = (value_loss, policy_loss)
        if return_returns:
            ret += (returns,)

        return ret
