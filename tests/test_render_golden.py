"""The render / record path against frames recorded from the REAL reference
(tests/golden/make_golden_render.py: SingleSnake.render single_snake.py:389-428, MultiSnake.render multi_snake.py:229-266).
CPU: the host-side frame assembly (tiling + Pillow resize) from the reference's own RGB batch; GPU: env.render() end to
end — the RGB batch comes from the observation kernels."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'render_frames.npz')


@pytest.fixture(scope='module')
def fx():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


def _args(a):
    return {'num_rows': int(a[0]), 'num_cols': int(a[1]), 'size': int(a[2])}


def _same_pillow(fx):
    import PIL
    return str(fx['pillow']) == PIL.__version__


@pytest.mark.parametrize('tag', ['single_tiled', 'single_one', 'multi'])
def test_frame_assembly_matches_the_reference(fx, tag):
    from wurm_amd._render import frame
    if not _same_pillow(fx):
        pytest.skip('Pillow differs from the one the frames were recorded with (resize filters changed across versions)')
    rgb = fx[tag + '_rgb']
    got = frame(rgb, rgb.shape[0], _args(fx[tag + '_args']))
    assert got.dtype == np.uint8 and np.array_equal(got, fx[tag + '_frame'])
    if tag == 'multi':
        assert np.array_equal(frame(rgb, rgb.shape[0], _args(fx['multi_args']), env=4), fx['multi_frame_env4'])


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['single_tiled', 'single_one'])
def test_single_snake_render(fx, tag):
    import torch
    from wurm_amd.envs import SingleSnake
    state = fx[tag + '_state']
    env = SingleSnake(num_envs=state.shape[0], size=12, manual_setup=True, device='cuda:0',
                      render_args=_args(fx[tag + '_args']))
    env.envs = torch.from_numpy(state).cuda()
    assert np.array_equal(env._get_rgb().cpu().numpy(), fx[tag + '_rgb'])
    if _same_pillow(fx):
        assert np.array_equal(env.render(mode='rgb_array'), fx[tag + '_frame'])


@pytest.mark.gpu
def test_multi_snake_render(fx):
    import torch
    from wurm_amd.envs import MultiSnake
    env = MultiSnake(num_envs=6, num_snakes=2, size=12, manual_setup=True, device='cuda:0',
                     render_args=_args(fx['multi_args']))
    env.foods, env.heads, env.bodies = (torch.from_numpy(fx['multi_' + k]).cuda() for k in ('foods', 'heads', 'bodies'))
    env.dones = torch.from_numpy(fx['multi_dones']).cuda().bool()
    env.boost_this_step = torch.from_numpy(fx['multi_boost']).cuda().bool()
    env.agent_colours = torch.from_numpy(fx['multi_colours']).cuda()
    assert np.array_equal(env._get_env_images().cpu().numpy(), fx['multi_rgb'])
    if _same_pillow(fx):
        assert np.array_equal(env.render(mode='rgb_array'), fx['multi_frame'])
        assert np.array_equal(env.render(mode='rgb_array', env=4), fx['multi_frame_env4'])
