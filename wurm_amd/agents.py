"""The feed-forward actor-critic of the reference's single-agent experiments (wurm/agents/feedforward.py:8-28:
`num_layers` blocks of Linear + ReLU, then a softmax action head and a state-value head) and the packing of its
weights for the fused acting kernel (`SingleSnake.policy_rollout`, include/wurm_hip.h: wurm_single_policy_rollout).

The module itself is plain torch (it is the learner's differentiable copy of the policy); the acting copy runs inside
the env kernel.  Only the architecture the kernel implements can be packed: 2 hidden layers of 64 units, 4 actions.
"""
import torch
from torch import nn
import torch.nn.functional as F


class FeedforwardAgent(nn.Module):
    """Same constructor, attributes and forward as the reference's `wurm.agents.FeedforwardAgent`."""

    def __init__(self, num_actions: int, num_layers: int, hidden_units: int, num_inputs: int = 4):
        super(FeedforwardAgent, self).__init__()
        self.num_layers = num_layers
        self.hidden_units = hidden_units
        self.num_actions = num_actions
        blocks = [nn.Sequential(nn.Linear(num_inputs, hidden_units), nn.ReLU())]
        for _ in range(num_layers - 1):
            blocks.append(nn.Sequential(nn.Linear(hidden_units, hidden_units), nn.ReLU()))
        self.feedforward = nn.Sequential(*blocks)
        self.action_head = nn.Linear(hidden_units, num_actions)
        self.value_head = nn.Linear(hidden_units, 1)

    def forward(self, x: torch.Tensor) -> (torch.Tensor, torch.Tensor):
        x = self.feedforward(x)
        return F.softmax(self.action_head(x), dim=-1), self.value_head(x)


def pack_policy_params(agent: FeedforwardAgent) -> torch.Tensor:
    """W1 (64,E) b1 (64) W2 (64,64) b2 (64) Wp (4,64) bp (4) Wv (64) bv (1) as one contiguous fp32 tensor on the
    agent's device — the `params` argument of wurm_single_policy_rollout."""
    if agent.num_layers != 2 or agent.hidden_units != 64 or agent.num_actions != 4:
        raise NotImplementedError('the fused acting kernel implements inputs -> 64 -> 64 -> {4 actions, 1 value}')
    l1, l2 = agent.feedforward[0][0], agent.feedforward[1][0]
    parts = [l1.weight, l1.bias, l2.weight, l2.bias, agent.action_head.weight, agent.action_head.bias,
             agent.value_head.weight, agent.value_head.bias]
    with torch.no_grad():
        return torch.cat([p.detach().to(torch.float32).reshape(-1) for p in parts]).contiguous()
