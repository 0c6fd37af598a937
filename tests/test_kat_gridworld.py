"""Known-answer tests of SimpleGridworld taken from the reference's test-suite (the boards,
action tapes and expected outcomes of tests/test_simple_gridworld.py in oscarknagg/wurm), table-driven, against
wurm_amd.envs.SimpleGridworld."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZE = 7
FOOD, AGENT = 0, 1

# (food cell, agent cell, action tape, expected agent cell after every step or None, reward per step or None,
#  step at which `done` must first be set or None)
CASES = {
    'moves':  ((1, 1), (3, 3), [0, 1, 2, 3, 2, 1], [(4, 3), (4, 2), (3, 2), (3, 3), (2, 3), (2, 2)], None, None),
    'eats':   ((1, 1), (2, 2), [0, 2, 2, 1], None, [0, 0, 0, 1], None),
    'leaves': ((1, 1), (3, 3), [0, 0, 0], None, None, 2),
}


def make_env(food, agent):
    from wurm_amd.envs import SimpleGridworld
    env = SimpleGridworld(num_envs=1, size=SIZE, start_location=(3, 3), manual_setup=True)
    env.envs[0, FOOD, food[0], food[1]] = 1
    env.envs[0, AGENT, agent[0], agent[1]] = 1
    return env


@pytest.mark.parametrize('case', sorted(CASES))
def test_known_answers(case):
    food, agent, tape, cells, rewards, done_at = CASES[case]
    env = make_env(food, agent)
    for i, a in enumerate(tape):
        action = torch.tensor([a], dtype=torch.long, device='cuda')
        obs, reward, done, info = env.step(action)
        assert action.item() == a                      # SimpleGridworld never rewrites actions
        if cells is not None:
            idx = int(env.envs[0, AGENT].flatten().argmax())
            assert (idx // SIZE, idx % SIZE) == cells[i]
        if rewards is not None:
            assert reward.item() == rewards[i]
        if done_at is not None:
            assert bool(done.item()) == (i >= done_at)
            assert bool(info['edge_collision'].item()) == (i >= done_at)
    if rewards is not None:                             # the eaten food respawned somewhere
        assert env.envs[0, FOOD].sum().item() == 1


def test_cfg1_random_rollout_keeps_invariants():
    """BASELINE cfg1 shape: 64 envs, 9x9, random actions, reset after every step."""
    from wurm_amd.envs import SimpleGridworld
    torch.manual_seed(0)
    env = SimpleGridworld(num_envs=64, size=9, start_location=(4, 4))
    tape = torch.randint(4, size=(200, 64), device='cuda')
    eaten = 0.0
    for a in tape:
        obs, reward, done, info = env.step(a)
        assert obs.shape == (64, 3, 9, 9) and reward.shape == (64, 1) and done.dtype == torch.bool
        eaten += reward.sum().item()
        env.reset(done)
        agent, food = env.envs[:, AGENT], env.envs[:, FOOD]
        assert torch.all(agent.sum(dim=(1, 2)) == 1) and torch.all(food.sum(dim=(1, 2)) == 1)
        assert torch.all((agent * food).sum(dim=(1, 2)) == 0)
    assert eaten > 0


def test_random_start_is_not_implemented():
    from wurm_amd.envs import SimpleGridworld
    with pytest.raises(NotImplementedError):
        SimpleGridworld(num_envs=2, size=9)
