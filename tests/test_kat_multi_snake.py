"""The reference's MultiSnake test-suite (tests/test_multi_snake_env.py in oscarknagg/wurm) re-expressed against
wurm_amd.envs.MultiSnake: same hand-built board, action tapes and assertions."""
import pytest
import torch

pytestmark = pytest.mark.gpu

size = 12
DEVICE = 'cuda'


@pytest.fixture(scope='module')
def api():
    from wurm_amd.envs import MultiSnake
    from wurm_amd.utils import determine_orientations
    return dict(MultiSnake=MultiSnake, determine_orientations=determine_orientations)


def get_test_env(api, num_envs=1, seed=1):
    """Two length-4 snakes per env (reference tests/test_multi_snake_env.py:21-47).  The reference seeds torch's
    global RNG once (`torch.random.manual_seed(1)`, :18); here the env's own seed is pinned."""
    env = api['MultiSnake'](num_envs=num_envs, num_snakes=2, size=size, manual_setup=True, seed=seed)
    for i in range(num_envs):
        env.heads[2 * i, 0, 5, 5] = 1
        for v, (y, x) in zip((4, 3, 2, 1), ((5, 5), (4, 5), (4, 4), (4, 3))):
            env.bodies[2 * i, 0, y, x] = v
        env.heads[2 * i + 1, 0, 8, 7] = 1
        for v, (y, x) in zip((4, 3, 2, 1), ((8, 7), (8, 8), (8, 9), (9, 9))):
            env.bodies[2 * i + 1, 0, y, x] = v
    _envs = torch.cat([env.foods.repeat_interleave(env.num_snakes, dim=0), env.heads, env.bodies], dim=1)
    env.orientations = api['determine_orientations'](_envs)
    assert env.orientations.tolist() == [2, 3] * num_envs
    return env


def tape(*rows):
    return {f'agent_{i}': torch.tensor(r).unsqueeze(1).long().to(DEVICE) for i, r in enumerate(rows)}


def run(env, all_actions, check=True, reset=False):
    for i in range(all_actions['agent_0'].shape[0]):
        actions = {agent: a[i] for agent, a in all_actions.items()}
        out = env.step(actions)
        if reset:
            env.reset(out[2]['__all__'])
        if check:
            env.check_consistency()
        yield (i,) + out


def head_pos(env, agent):
    idx = env.heads[agent, 0].flatten().argmax()
    return [int(idx // size), int(idx % size)]


def test_random_actions(api):
    env = api['MultiSnake'](num_envs=100, num_snakes=2, size=size)
    env.check_consistency()
    acts = {f'agent_{i}': torch.randint(4, size=(100, 100)).long().to(DEVICE) for i in range(2)}
    for i in range(100):
        obs, reward, done, info = env.step({k: v[i] for k, v in acts.items()})
        env.reset(done['__all__'])
        env.check_consistency()


def test_random_actions_with_boost(api):
    env = api['MultiSnake'](num_envs=256, num_snakes=4, size=25, boost=True, respawn_mode='any',
                            food_mode='random_rate', boost_cost_prob=0.25, observation_mode='partial_5',
                            food_on_death_prob=0.33, food_rate=2.5e-4)
    env.check_consistency()
    acts = {f'agent_{i}': torch.randint(8, size=(200, 256)).long().to(DEVICE) for i in range(4)}
    for i in range(200):
        obs, reward, done, info = env.step({k: v[i] for k, v in acts.items()})
        assert obs['agent_3'].shape == (256, 3, 11, 11)
        env.reset(done['__all__'], return_observations=False)
        env.check_consistency()


def test_basic_movement(api):
    env = get_test_env(api)
    env.foods[0, 0, 1, 1] = 1
    expected = [[[5, 4], [4, 4], [4, 3], [4, 2], [5, 2], [5, 3]],
                [[9, 7], [9, 6], [9, 5], [8, 5], [8, 4], [9, 4]]]
    for i, obs, rewards, dones, info in run(env, tape([1, 2, 1, 1, 0, 3], [0, 1, 3, 2, 1, 0])):
        for a in range(2):
            assert head_pos(env, a) == expected[a][i]
        assert not any(bool(d.any()) for d in dones.values())


def test_edge_collision(api):
    env = get_test_env(api)
    env.food_on_death_prob = 1
    env.foods[0, 0, 1, 1] = 1
    for i, obs, rewards, dones, info in run(env, tape([1, 1, 1, 1, 1], [0, 2, 2, 6, 2])):
        if i == 4:
            assert rewards['agent_0'].item() == env.reward_on_death
        if i == 2:
            assert rewards['agent_1'].item() == env.reward_on_death
        assert dones['agent_0'].item() == (1 if i >= 4 else 0)
        assert dones['agent_1'].item() == (1 if i >= 2 else 0)


def test_self_collision(api):
    env = get_test_env(api)
    env.food_on_death_prob = 1
    env.foods[0, 0, 4, 3] = 1
    for i, obs, rewards, dones, info in run(env, tape([1, 2, 1, 1, 0, 3, 2, 0], [0, 1, 3, 2, 1, 0, 0, 1])):
        assert dones['agent_0'].item() == (1 if i >= 6 else 0)


def test_other_snake_collision(api):
    env = get_test_env(api)
    env.foods[0, 0, 1, 1] = 1
    env.food_on_death_prob = 1
    for i, obs, rewards, dones, info in run(env, tape([1, 2, 3, 3, 3, 3, 3, 2], [1, 2, 2, 2, 2, 2, 2, 2])):
        assert dones['agent_1'].item() == (1 if i >= 4 else 0)
    assert env.foods[:, 0].sum().item() >= 2  # food was created on death


def test_eat_food(api):
    env = get_test_env(api, num_envs=1)
    env.foods[:, 0, 9, 7] = 1
    for i, obs, rewards, dones, info in run(env, tape([1, 2, 1, 1, 0, 3], [0, 1, 3, 2, 1, 0])):
        assert rewards['agent_1'].item() == (1 if i == 0 else 0)
        assert not any(bool(d.any()) for d in dones.values())
    sizes = env.bodies.view(1, 2, -1).max(dim=2)[0].long()
    assert sizes.tolist() == [[4, 5]]
    assert env.foods[0, 0, 9, 7].item() == 0
    assert env.foods.sum().item() == 1


def test_create_envs(api):
    env = api['MultiSnake'](num_envs=512, num_snakes=2, size=size)
    env.check_consistency()
    _envs = torch.cat([env.foods.repeat_interleave(env.num_snakes, dim=0), env.heads, env.bodies], dim=1)
    assert torch.equal(env.orientations, api['determine_orientations'](_envs))


def test_reset(api):
    env = get_test_env(api, num_envs=1)
    env.foods[:, 0, 1, 1] = 1
    for _ in run(env, tape([1, 2, 3, 3, 3, 3, 3, 3, 3], [0, 1, 2, 2, 2, 2, 2, 2, 2]), reset=True):
        pass
    assert torch.all(env.bodies.view(1, 2, -1).max(dim=-1)[0] == env.initial_snake_length)


def test_agent_observations(api):
    env = get_test_env(api, num_envs=1)
    env.foods[:, 0, 1, 1] = 1
    obs_0, obs_1 = env._observe_agent(0), env._observe_agent(1)
    half_self, half_other = env.self_colour.float() / 2, env.other_colour.float() / 2
    assert torch.allclose(obs_0[0, :, 4, 5] * 255, half_self)
    assert torch.allclose(obs_0[0, :, 8, 8] * 255, half_other)
    assert torch.allclose(obs_1[0, :, 4, 5] * 255, half_other)
    assert torch.allclose(obs_1[0, :, 8, 8] * 255, half_self)


def test_boost_through_food(api):
    env = get_test_env(api, num_envs=1)
    env.boost = True
    env.foods[:, 0, 6, 5] = 1
    env.boost_cost_prob = 0
    for i, obs, rewards, dones, info in run(env, tape([4, 1, 2], [0, 1, 3]), reset=True):
        if i == 0:
            assert rewards['agent_0'].item() == 1


def test_boost_leaves_food(api):
    env = get_test_env(api, num_envs=1)
    env.boost = True
    env.boost_cost_prob = 1
    for i, obs, rewards, dones, info in run(env, tape([4, 1, 2], [0, 1, 3]), reset=True):
        if i == 0:
            assert rewards['agent_0'].item() == -1
    assert env.foods[0, 0, 4, 4].item() == 1


def test_cant_boost_until_size_4(api):
    env = api['MultiSnake'](num_envs=1, num_snakes=2, size=size, manual_setup=True, boost=True)
    env.foods[:, 0, 1, 1] = 1
    env.heads[0, 0, 5, 5] = 1
    for v, (y, x) in zip((3, 2, 1), ((5, 5), (4, 5), (4, 4))):
        env.bodies[0, 0, y, x] = v
    env.heads[1, 0, 8, 7] = 1
    for v, (y, x) in zip((3, 2, 1), ((8, 7), (8, 8), (8, 9))):
        env.bodies[1, 0, y, x] = v
    _envs = torch.cat([env.foods.repeat_interleave(env.num_snakes, dim=0), env.heads, env.bodies], dim=1)
    env.orientations = api['determine_orientations'](_envs)
    expected = [[6, 5], [6, 4], [5, 4]]
    for i, obs, rewards, dones, info in run(env, tape([4, 1, 2], [0, 1, 3]), reset=True):
        assert head_pos(env, 0) == expected[i]
        assert not info['boost_0'].item()


def test_boost_cost(api):
    env = get_test_env(api, num_envs=1)
    env.boost = True
    env.boost_cost_prob = 1
    env.foods[:, 0, 1, 1] = 1
    for i, obs, rewards, dones, info in run(env, tape([4, 1, 2], [0, 1, 3]), reset=True):
        sizes = env.bodies.view(1, 2, -1).max(dim=2)[0].long()
        assert sizes.tolist() == [[3, 4]]
        if i == 0:
            assert rewards['agent_0'].item() == -1


def test_many_snakes(api):
    env = api['MultiSnake'](num_envs=50, num_snakes=4, size=size, boost=True)
    env.check_consistency()
    acts = {f'agent_{i}': torch.randint(8, size=(10, 50)).long().to(DEVICE) for i in range(4)}
    for i in range(10):
        obs, reward, done, info = env.step({k: v[i] for k, v in acts.items()})
        env.reset(done['__all__'])
        env.check_consistency()


def test_boost_rendering(api):
    pytest.importorskip('PIL')
    env = get_test_env(api, num_envs=1)
    env.boost = True
    env.foods[:, 0, 1, 5] = 1
    colours = []
    for i, obs, rewards, dones, info in run(env, tape([4, 1, 2], [0, 1, 3]), check=False):
        env.reset(dones['__all__'], return_observations=False)
        img = env._get_env_images()
        colours.append(img[0, :, 5, 5].float().norm().item())  # a body cell of agent_0 (the reference samples
        # pixel (127,127) of the 256-px render = cell (5,5))
        env.check_consistency()
    assert colours[0] > colours[1]  # the body appears brighter while boosting
    assert env.render(mode='rgb_array').shape == (256, 256, 3)


def test_respawn_mode_any(api):
    env = get_test_env(api)
    env.respawn_mode = 'any'
    for i in range(2, 9, 2):
        for j in range(2, 9, 2):
            env.foods[0, 0, i, j] = 1  # food blocks every respawn location
    for _ in run(env, tape([1, 1, 1, 1, 2, 2, 2, 3], [0, 1, 0, 0, 0, 0, 0, 1]), reset=True):
        pass


def test_partial_observations(api):
    env = api['MultiSnake'](num_envs=256, num_snakes=4, size=25, boost=True, observation_mode='partial_5')
    env.check_consistency()
    observations = env._observe('partial_5')
    assert list(observations) == [f'agent_{i}' for i in range(4)]
    for v in observations.values():
        assert v.shape == (256, 3, 11, 11) and v.dtype == torch.float32
        centre = v[:, :, 5, 5]  # the window is centred on the agent's own head: coloured, not background white
        assert torch.all((centre < 1).any(dim=1))


def test_argument_errors(api):
    env = api['MultiSnake'](num_envs=3, num_snakes=2, size=size)
    ok = torch.zeros(3, dtype=torch.long, device=DEVICE)
    with pytest.raises(RuntimeError):
        env.step({'agent_0': ok})
    with pytest.raises(TypeError):
        env.step({'agent_0': ok, 'agent_1': ok.float()})
    with pytest.raises(RuntimeError):
        env.step({'agent_0': ok, 'agent_1': torch.zeros(4, dtype=torch.long, device=DEVICE)})
    with pytest.raises(ValueError):
        api['MultiSnake'](num_envs=3, num_snakes=2, size=size, agent_colours='stripes')


def test_output_packaging(api):
    env = api['MultiSnake'](num_envs=5, num_snakes=3, size=size)
    acts = {f'agent_{i}': torch.randint(8, (5,), device=DEVICE) for i in range(3)}
    obs, rewards, dones, info = env.step(acts)
    assert list(obs) == ['agent_0', 'agent_1', 'agent_2'] and obs['agent_0'].shape == (5, 3, size, size)
    assert set(dones) == {'agent_0', 'agent_1', 'agent_2', '__all__'} and dones['__all__'].shape == (5,)
    assert rewards['agent_2'].shape == (5,) and rewards['agent_2'].dtype == torch.float32
    for key in ('snake_collision_', 'edge_collision_', 'food_', 'boost_', 'size_'):
        for i in range(3):
            assert info[f'{key}{i}'].shape == (5,)
    assert torch.equal(torch.stack([dones[f'agent_{i}'] for i in range(3)], dim=1).flatten(), env.dones)


def test_rollout_equals_python_loop(api):
    N, K, T = 64, 4, 60
    kw = dict(num_envs=N, num_snakes=K, size=25, respawn_mode='any', food_mode='random_rate', boost_cost_prob=0.25,
              observation_mode='partial_5', food_on_death_prob=0.33, food_rate=2.5e-3, seed=77)
    a, b = api['MultiSnake'](**kw), api['MultiSnake'](**kw)
    assert torch.equal(a.bodies, b.bodies) and torch.equal(a.agent_colours, b.agent_colours)
    actions = torch.randint(8, size=(T, K, N), device=DEVICE)
    out = b.rollout(actions)
    for t in range(T):
        obs, rewards, dones, info = a.step({f'agent_{i}': actions[t, i] for i in range(K)})
        for i in range(K):
            assert torch.equal(obs[f'agent_{i}'], out['observations'][t, i]), (t, i)
            assert torch.equal(rewards[f'agent_{i}'], out['rewards'][t, i])
            assert torch.equal(dones[f'agent_{i}'], out['dones'][t, i])
            assert torch.equal(info[f'size_{i}'], out['size'][t, i])
        assert torch.equal(dones['__all__'], out['all_done'][t])
        a.reset(dones['__all__'], return_observations=False)
    for name in ('foods', 'heads', 'bodies', 'dones', 'orientations', 'agent_colours', 'boost_this_step'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    b.check_consistency()


def test_half_dtype(api):
    """`dtype=torch.half` (reference :64): what the agents receive is half, computed as the reference's
    `.to(dtype) / 255` (:281) would; the reference's own half path cannot step (the einsum at :640 mixes half and float
    and raises), so the check is against the fp32 env: same trajectories, outputs equal to its outputs rounded once."""
    kw = dict(num_envs=6, num_snakes=3, size=12, seed=5, observation_mode='partial_3', food_mode='random_rate')
    a, b = api['MultiSnake'](**kw), api['MultiSnake'](dtype=torch.half, **kw)
    assert b.foods.dtype == torch.float32  # the state stays fp32 (docstring)
    tape = torch.randint(8, (12, 3, 6), device=DEVICE)
    for t in range(8):
        acts = {f'agent_{i}': tape[t, i] for i in range(3)}
        oa, ra, da, ia = a.step(acts)
        ob, rb, db, ib = b.step(acts)
        for i in range(3):
            assert ob[f'agent_{i}'].dtype == torch.half and torch.equal(ob[f'agent_{i}'], oa[f'agent_{i}'].half())
            assert ib[f'size_{i}'].dtype == torch.half and torch.equal(ib[f'size_{i}'], ia[f'size_{i}'].half())
            assert ib[f'food_{i}'].dtype == torch.half and torch.equal(ib[f'food_{i}'], ia[f'food_{i}'].half())
            assert rb[f'agent_{i}'].dtype == torch.float32 and torch.equal(rb[f'agent_{i}'], ra[f'agent_{i}'])
            assert torch.equal(db[f'agent_{i}'], da[f'agent_{i}'])
        ba, bb = a.reset(da['__all__']), b.reset(db['__all__'])
        assert all(bb[k].dtype == torch.half and torch.equal(bb[k], ba[k].half()) for k in ba)
    ra, rb = a.rollout(tape[8:]), b.rollout(tape[8:])
    assert rb['observations'].dtype == torch.half and torch.equal(rb['observations'], ra['observations'].half())
    assert torch.equal(rb['size'], ra['size'].half()) and torch.equal(rb['rewards'], ra['rewards'])
    assert torch.equal(a.bodies, b.bodies)
    with pytest.raises(NotImplementedError):
        api['MultiSnake'](num_envs=2, num_snakes=2, size=12, dtype=torch.double)
