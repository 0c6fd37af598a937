#!/usr/bin/env python3
"""The single-agent A2C loop of the reference (experiments/main.py:194-247, which is broken at the reference's commit:
`A2C(model, gamma=...)` vs wurm/rl/a2c.py:18-24) running unchanged in structure against wurm_amd: policy -> Categorical
-> env.step -> trajectory store -> env.reset -> every `update_steps`: bootstrap, A2C loss, Adam.

Everything stays on the GPU; per step the host only enqueues kernels.  The statistics of main.py:252-274 are
accumulated on the device by `wurm_single_stats` and read back once per log interval.

    python examples/a2c_loop.py --num-envs 512 --size 9 --observation partial_2 --steps 2000
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch import nn  # noqa: E402
from torch.distributions import Categorical  # noqa: E402

from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402
from wurm_amd.rl import A2C, TrajectoryStore  # noqa: E402


class FeedforwardAgent(nn.Module):
    """MLP policy with the layer sizes of the reference's wurm/agents/feedforward.py:8-28 (inputs -> 64 -> 64 -> 4 / 1)."""

    def __init__(self, num_inputs: int, num_actions: int = 4, hidden: int = 64):
        super().__init__()
        self.body = nn.Sequential(nn.Linear(num_inputs, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU())
        self.policy = nn.Linear(hidden, num_actions)
        self.value = nn.Linear(hidden, 1)

    def forward(self, x):
        h = self.body(x.flatten(1))
        return torch.softmax(self.policy(h), dim=-1), self.value(h)


def run(num_envs=512, size=9, observation='partial_2', steps=2000, update_steps=5, gamma=0.99, lr=1e-3, entropy=0.01,
        log_interval=500, seed=0, device='cuda', verbose=True):
    torch.manual_seed(seed)
    env = SingleSnake(num_envs=num_envs, size=size, observation_mode=observation, device=device)
    state = env.reset()                                                     # main.py:195
    model = FeedforwardAgent(state[0].numel()).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    a2c = A2C(gamma=gamma)
    trajectories = TrajectoryStore(capacity=update_steps)
    stats = torch.zeros(5, dtype=torch.float64, device=device)
    history, t0, last = [], time.perf_counter(), 0
    for i_step in range(1, steps + 1):                                      # main.py:196
        probs, state_value = model(state)                                   # :207
        dist = Categorical(probs)
        ent = dist.entropy().mean()
        action = dist.sample().clone().long()                               # :210
        state, reward, done, info = env.step(action)                        # :212 (action sanitised in place)
        trajectories.append(action=action, log_prob=dist.log_prob(action).unsqueeze(-1), value=state_value,
                            reward=reward, done=done, entropy=ent)          # :217-225 (log_prob of the sanitised action)
        _lib.check(_lib.lib().wurm_single_stats(
            _lib.ptr(env.envs), _lib.ptr(reward), _lib.ptr(done), _lib.ptr(info['self_collision']),
            _lib.ptr(info['edge_collision']), _lib.ptr(stats), _lib.i64(num_envs), size,
            _lib.stream_ptr(env.device.index)), 'stats')                    # :252-274 without a sync
        env.reset(done, return_observations=False)                          # :227 (its observation is discarded)
        if i_step % update_steps == 0:                                      # :232-247
            with torch.no_grad():
                _, bootstrap_values = model(state)
            value_loss, policy_loss = a2c.loss(bootstrap_values, trajectories.rewards, trajectories.values,
                                               trajectories.log_probs, trajectories.dones)
            entropy_loss = -trajectories.entropies.mean()
            optimizer.zero_grad()
            loss = value_loss + policy_loss + entropy * entropy_loss
            loss.backward()
            nn.utils.clip_grad_norm_(model.parameters(), 0.5)
            optimizer.step()
            trajectories.clear()
        if i_step % log_interval == 0 or i_step == steps:
            s = stats.cpu().tolist()
            stats.zero_()
            n = (i_step - last) * num_envs
            dt, last = time.perf_counter() - t0, i_step
            row = dict(step=i_step, env_steps_per_s=i_step * num_envs / dt, done_rate=s[0] / n, reward_rate=s[1] / n,
                       edge_rate=s[2] / n, self_rate=s[3] / n, mean_length=s[4] / n, loss=float(loss))
            history.append(row)
            if verbose:
                print(' '.join(f'{k}={v:.4g}' for k, v in row.items()))
    return history


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--num-envs', type=int, default=512)
    ap.add_argument('--size', type=int, default=9)
    ap.add_argument('--observation', default='partial_2')
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--update-steps', type=int, default=5)
    ap.add_argument('--lr', type=float, default=1e-3)
    args = ap.parse_args()
    run(args.num_envs, args.size, args.observation, args.steps, args.update_steps, lr=args.lr)
