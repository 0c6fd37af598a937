"""Host-side frame assembly for `render(mode='rgb_array')` (reference: single_snake.py:389-428, multi_snake.py:229-266).
Not on the step path: takes the int16 RGB batch an observation kernel produced and tiles / upsamples it on the CPU."""
import numpy as np


def frame(rgb_nchw, num_envs: int, render_args: dict, env: int = None) -> np.ndarray:
    """rgb_nchw: (N,3,S,S) integer array.  One env (num_envs == 1 or `env` given) fills the frame; otherwise the first
    num_rows x num_cols envs are tiled row-major.  The result is resized to render_args['size'] pixels per tile."""
    from PIL import Image

    imgs = np.asarray(rgb_nchw).astype(np.uint8).transpose(0, 2, 3, 1)  # N,S,S,3
    S = imgs.shape[1]
    if num_envs == 1 or env is not None:
        rows = cols = 1
        canvas = imgs[env or 0]
    else:
        rows, cols = render_args['num_rows'], render_args['num_cols']
        tiles = imgs[:rows * cols].reshape(rows, cols, S, S, 3)
        canvas = tiles.transpose(0, 2, 1, 3, 4).reshape(rows * S, cols * S, 3)
    px = render_args['size']
    return np.array(Image.fromarray(np.ascontiguousarray(canvas)).resize((px * cols, px * rows)))
