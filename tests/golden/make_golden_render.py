"""Golden frames for the render / record path, recorded from the REAL reference in the build
container: `env.render(mode='rgb_array')` of wurm.envs.SingleSnake (single_snake.py:389-428) and MultiSnake
(multi_snake.py:229-266, incl. a boosting snake) on states taken from the step fixtures, for a single env, for the
tiled view and for MultiSnake's `env=` argument.  Stored: the state, the int16 RGB batch the reference rendered from
(`_get_rgb` / `_get_env_images`) and the uint8 frame.  Data only; see make_golden.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

SingleSnake, SimpleGridworld, MultiSnake, ref_utils = ref_shim.import_reference()


def main():
    out = {}
    z = np.load(os.path.join(HERE, 'single_s12_default.npz'))
    state = z['state_step'][40].astype(np.float32)          # (32,3,12,12) mid-tape: bent bodies, some dead snakes
    for tag, n, args in (('single_tiled', 6, {'num_rows': 2, 'num_cols': 3, 'size': 48}),
                         ('single_one', 1, {'num_rows': 1, 'num_cols': 1, 'size': 60})):
        env = SingleSnake(num_envs=n, size=12, manual_setup=True, device='cpu', render_args=args)
        env.envs = torch.from_numpy(state[:n].copy())
        out[tag + '_state'] = state[:n]
        out[tag + '_rgb'] = env._get_rgb().numpy().astype(np.int16)
        out[tag + '_frame'] = env.render(mode='rgb_array')
        out[tag + '_args'] = np.asarray([args['num_rows'], args['num_cols'], args['size']])

    m = np.load(os.path.join(HERE, 'multi_k2_s12_default.npz'))
    t, N, K, S = 30, 6, 2, 12
    args = {'num_rows': 2, 'num_cols': 3, 'size': 36}
    env = MultiSnake(num_envs=N, num_snakes=K, size=S, manual_setup=True, device='cpu', render_args=args)
    env.foods = torch.from_numpy(m['step_foods'][t][:N].astype(np.float32))
    env.heads = torch.from_numpy(m['step_heads'][t][:N * K].astype(np.float32))
    env.bodies = torch.from_numpy(m['step_bodies'][t][:N * K].astype(np.float32))
    env.dones = torch.from_numpy(m['step_dones'][t][:N * K])
    boost = m['step_boost_this_step'][t][:N * K].copy()
    boost[0] = 1                                            # make sure a boosting snake is in the picture
    env.boost_this_step = torch.from_numpy(boost)
    env.agent_colours = torch.from_numpy(m['step_colours'][t][:N * K])
    for k in ('foods', 'heads', 'bodies'):
        out['multi_' + k] = getattr(env, k).numpy()
    out['multi_dones'], out['multi_boost'], out['multi_colours'] = env.dones.numpy(), boost, env.agent_colours.numpy()
    out['multi_rgb'] = env._get_env_images().numpy().astype(np.int16)
    out['multi_frame'] = env.render(mode='rgb_array')
    out['multi_frame_env4'] = env.render(mode='rgb_array', env=4)
    out['multi_args'] = np.asarray([2, 3, 36])
    import PIL
    out['pillow'] = np.asarray(PIL.__version__)
    np.savez_compressed(os.path.join(HERE, 'render_frames.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
