#!/usr/bin/env python3
"""The single-agent A2C experiment of the reference (experiments/main.py:194-247) with the ACTING half fused into the
env kernel: `env.policy_rollout(params, state, update_steps)` runs `update_steps` iterations of
policy -> Categorical sample -> env.step -> env.reset on the GPU in one launch, and the learner recomputes
probabilities and values WITH gradients from the returned observations in one batched forward pass (the weights do not
change inside the window, so these are the values the actor used — to fp32 rounding), then takes the reference's A2C
step.  Per update the host issues ~25 kernel launches instead of ~40 per env step.

    python examples/a2c_fused_actor.py --num-envs 512 --steps 20000
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch import nn  # noqa: E402
from torch.distributions import Categorical  # noqa: E402

from wurm_amd.agents import FeedforwardAgent, pack_policy_params  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402
from wurm_amd.rl import A2C  # noqa: E402


def run(num_envs=512, size=9, observation='partial_2', steps=20000, update_steps=5, gamma=0.99, lr=1e-3, entropy=0.01,
        log_interval=2000, seed=0, device='cuda', verbose=True):
    torch.manual_seed(seed)
    env = SingleSnake(num_envs=num_envs, size=size, observation_mode=observation, device=device, seed=seed)
    state = env.reset()                                                     # main.py:195
    model = FeedforwardAgent(num_actions=4, num_layers=2, hidden_units=64, num_inputs=state[0].numel()).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=lr)
    a2c = A2C(gamma=gamma)
    totals = torch.zeros(2, dtype=torch.float64, device=device)             # rewards, dones since the last log line
    history, t0, last, loss = [], time.perf_counter(), 0, torch.zeros(())
    for i_step in range(update_steps, steps + 1, update_steps):
        out = env.policy_rollout(pack_policy_params(model), state, update_steps, check=False)   # :207-227, fused
        inputs = torch.cat([state.unsqueeze(0), out['observations'][:-1]]).flatten(2)           # what the actor saw
        state = out['state']
        probs, values = model(inputs)                                                           # (T,N,4), (T,N,1)
        dist = Categorical(probs)                                                                # :208
        log_probs = dist.log_prob(out['actions']).unsqueeze(-1)                                 # :220 (sanitised action)
        entropies = dist.entropy().mean(-1)                                                     # :209, one per step
        with torch.no_grad():
            _, bootstrap_values = model(state.flatten(1))                                       # :233-234
        value_loss, policy_loss = a2c.loss(bootstrap_values, out['rewards'].unsqueeze(-1), values, log_probs,
                                           out['dones'].unsqueeze(-1))                          # :236-237
        loss = value_loss + policy_loss - entropy * entropies.mean()                            # :239-242
        optimizer.zero_grad()
        loss.backward()
        nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        optimizer.step()
        totals += torch.stack([out['rewards'].sum(dtype=torch.float64), out['dones'].sum(dtype=torch.float64)])
        if i_step % log_interval < update_steps or i_step + update_steps > steps:
            s = totals.cpu().tolist()
            totals.zero_()
            n = (i_step - last) * num_envs
            dt, last = time.perf_counter() - t0, i_step
            row = dict(step=i_step, env_steps_per_s=i_step * num_envs / dt, reward_rate=s[0] / n, done_rate=s[1] / n,
                       loss=float(loss))
            history.append(row)
            if verbose:
                print(' '.join(f'{k}={v:.4g}' for k, v in row.items()))
    return history


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--num-envs', type=int, default=512)
    ap.add_argument('--size', type=int, default=9)
    ap.add_argument('--observation', default='partial_2')
    ap.add_argument('--steps', type=int, default=20000)
    ap.add_argument('--update-steps', type=int, default=5)
    ap.add_argument('--lr', type=float, default=1e-3)
    args = ap.parse_args()
    run(args.num_envs, args.size, args.observation, args.steps, args.update_steps, lr=args.lr)
