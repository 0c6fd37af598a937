"""Golden vectors for the acting policy, recorded from the REAL reference (wurm/agents/feedforward.py:8-28) in the
build container: the weights of `wurm.agents.FeedforwardAgent(num_actions=4, num_layers=2, hidden_units=64,
num_inputs=E)` as torch initialises them (packed in the order of include/wurm_hip.h: wurm_single_policy_rollout), a
batch of real observations (what SingleSnake returns in 'partial_n' mode: values 0, 1, 127/255) and the reference's
`probs, values = model(obs)`.  Data only; see make_golden.py.  Run: python tests/golden/make_golden_policy.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_shim  # noqa: E402

ref_shim.install()
from wurm.agents import FeedforwardAgent  # noqa: E402  (the reference's class)
from wurm.envs import SingleSnake  # noqa: E402        (the reference's env: real observations)


def record(name, n, size, seed, M=96, scale=1.0):
    E = 3 * (2 * n + 1) ** 2
    torch.manual_seed(seed)
    model = FeedforwardAgent(num_actions=4, num_layers=2, hidden_units=64, num_inputs=E)
    if scale != 1.0:  # sharper action distributions than a fresh initialisation gives
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(scale)
    env = SingleSnake(num_envs=M, size=size, observation_mode=f'partial_{n}', device='cpu')
    obs = env.reset()
    for _ in range(6):  # a few steps so that bodies bend and food moves
        obs, _, done, _ = env.step(torch.randint(4, (M,)))
        env.reset(done)
    obs = obs.reshape(M, E).float()
    with torch.no_grad():
        probs, values = model(obs)
    sd = model.state_dict()
    order = ['feedforward.0.0.weight', 'feedforward.0.0.bias', 'feedforward.1.0.weight', 'feedforward.1.0.bias',
             'action_head.weight', 'action_head.bias', 'value_head.weight', 'value_head.bias']
    params = np.concatenate([sd[k].numpy().reshape(-1) for k in order]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), params=params, obs=obs.numpy(), probs=probs.numpy(),
                        values=values.numpy(), meta=np.asarray([M, E, n, size]))
    print(name, params.shape, obs.shape, float(probs.min()), float(probs.max()))


if __name__ == '__main__':
    record('policy_ff_n2_s9', 2, 9, seed=11)    # the headline shape: 75 inputs
    record('policy_ff_n1_s10', 1, 10, seed=12)  # 27 inputs
    record('policy_ff_n3_s11', 3, 11, seed=13)  # 147 inputs
    record('policy_ff_n2_s9_sharp', 2, 9, seed=14, scale=4.0)
