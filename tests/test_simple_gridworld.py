"""The reference's SimpleGridworld tests (tests/test_simple_gridworld.py in oscarknagg/wurm) re-expressed against
wurm_amd.envs.SimpleGridworld."""
import pytest
import torch

pytestmark = pytest.mark.gpu

size = 7
DEVICE = 'cuda'
FOOD_CHANNEL, HEAD_CHANNEL = 0, 1


@pytest.fixture(scope='module')
def SimpleGridworld():
    from wurm_amd.envs import SimpleGridworld
    return SimpleGridworld


def _head(env):
    idx = env.envs[0, HEAD_CHANNEL].flatten().argmax()
    return torch.Tensor([idx // size, idx % size])


def test_basic_movement(SimpleGridworld):
    env = SimpleGridworld(num_envs=1, size=size, start_location=(3, 3), manual_setup=True)
    env.envs[0, FOOD_CHANNEL, 1, 1] = 1
    env.envs[0, HEAD_CHANNEL, 3, 3] = 1
    actions = torch.Tensor([0, 1, 2, 3, 2, 1]).unsqueeze(1).long().to(DEVICE)
    expected = torch.Tensor([[4, 3], [4, 2], [3, 2], [3, 3], [2, 3], [2, 2]])
    for i, a in enumerate(actions):
        observations, reward, done, info = env.step(a)
        assert torch.equal(_head(env), expected[i])


def test_eat_food(SimpleGridworld):
    env = SimpleGridworld(num_envs=1, size=size, start_location=(3, 3), manual_setup=True)
    env.envs[0, FOOD_CHANNEL, 1, 1] = 1
    env.envs[0, HEAD_CHANNEL, 2, 2] = 1
    actions = torch.Tensor([0, 2, 2, 1]).unsqueeze(1).long().to(DEVICE)
    rewards = []
    for a in actions:
        observations, reward, done, info = env.step(a)
        rewards.append(reward.item())
    assert rewards == [0, 0, 0, 1]
    assert env.envs[0, FOOD_CHANNEL].sum().item() == 1  # respawned


def test_edge_collision(SimpleGridworld):
    env = SimpleGridworld(num_envs=1, size=size, start_location=(3, 3), manual_setup=True)
    env.envs[0, FOOD_CHANNEL, 1, 1] = 1
    env.envs[0, HEAD_CHANNEL, 3, 3] = 1
    actions = torch.Tensor([0, 0, 0, 0]).unsqueeze(1).long().to(DEVICE)
    for i, a in enumerate(actions):
        observations, reward, done, info = env.step(a)
        if i == 2:
            assert done.item()
            break
        else:
            assert not done.item()


def test_random_rollout_and_reset(SimpleGridworld):
    """BASELINE cfg1 shape: 64 envs, 9x9, random actions."""
    torch.manual_seed(0)
    env = SimpleGridworld(num_envs=64, size=9, start_location=(4, 4))
    actions = torch.randint(4, size=(200, 64)).long().to(DEVICE)
    for a in actions:
        obs, reward, done, info = env.step(a)
        assert obs.shape == (64, 3, 9, 9)
        env.reset(done)
        assert torch.all(env.envs[:, HEAD_CHANNEL].sum(dim=(1, 2)) == 1)
        assert torch.all(env.envs[:, FOOD_CHANNEL].sum(dim=(1, 2)) == 1)
        assert torch.all((env.envs[:, HEAD_CHANNEL] * env.envs[:, FOOD_CHANNEL]).sum(dim=(1, 2)) == 0)


def test_missing_start_location(SimpleGridworld):
    with pytest.raises(NotImplementedError):
        SimpleGridworld(num_envs=2, size=9)
