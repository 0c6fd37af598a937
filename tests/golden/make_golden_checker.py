"""Pins the invariant checkers on FAILING states: recorded states are corrupted in the ways the
reference's checks look for, and for every variant the REAL reference is asked whether — and with which message — it
raises (wurm/utils.py:113-178 `snake_consistency` / `env_consistency`; wurm/envs/multi_snake.py:733-769
`MultiSnake.check_consistency`).  The fixture holds DATA only: the corrupted states and the reference's verdicts.

Run only in the build container (where /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_checker.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

SingleSnake, SimpleGridworld, MultiSnake, ref_utils = ref_shim.import_reference()


def verdict(fn):
    try:
        fn()
        return ''
    except RuntimeError as e:
        return str(e)


def single_variants(S, seed):
    torch.manual_seed(seed)
    env = SingleSnake(num_envs=1, size=S, device='cpu')
    # grow the snake a little so that removing / duplicating body values has room
    base = env.envs.clone()
    out = []

    def add(name, e):
        out.append((name, e.clone()))

    def body_cell(e, v):
        ys, xs = np.nonzero(e[0, 2].numpy() == v)
        return int(ys[0]), int(xs[0])

    L = int(base[0, 2].max())
    hy, hx = body_cell(base, L)
    ty, tx = body_cell(base, 1)
    fy, fx = [int(v[0]) for v in np.nonzero(base[0, 0].numpy() > 0)]
    free = [(y, x) for y in range(1, S - 1) for x in range(1, S - 1) if float(base[0, :, y, x].sum()) == 0]

    add('valid', base)
    e = base.clone(); e[0, 1, free[0][0], free[0][1]] = 1; add('two_heads', e)
    e = base.clone(); e[0, 1] = 0; add('no_head', e)
    e = base.clone(); e[0, 0, fy, fx] = 2; add('food_value_2', e)
    e = base.clone(); e[0, 0, free[1][0], free[1][1]] = 0.5; add('food_value_half', e)
    e = base.clone(); e[0, 2] = 0; add('no_body', e)
    e = base.clone(); e[0, 1] = 0; e[0, 2] = 0; add('no_snake_at_all', e)
    my, mx = body_cell(base, 2)
    e = base.clone(); e[0, 2, my, mx] = 0; add('body_value_2_missing', e)
    e = base.clone(); e[0, 1] = 0; e[0, 1, ty, tx] = 1; add('head_on_the_tail', e)
    e = base.clone(); e[0, 2, ty, tx] = 0; e[0, 2, my, mx] = 1; e[0, 2, hy, hx] = 2; add('length_2_snake', e)
    e = base.clone(); e[0, 0] = 0; e[0, 0, hy, hx] = 1; add('food_under_the_head', e)
    e = base.clone(); e[0, 0, free[2][0], free[2][1]] = 1; add('two_foods', e)
    e = base.clone(); e[0, 0] = 0; add('no_food', e)
    e = base.clone(); e[0, 2, free[3][0], free[3][1]] = 2; add('body_value_2_twice', e)
    e = base.clone(); e[0, 2, ty, tx] = 0; e[0, 2, my, mx] = 3; add('values_3_3_sum_is_triangular', e)
    e = base.clone(); e[0, 2, hy, hx] = L + 1; add('head_value_too_large', e)
    e = base.clone(); e[0, 0, fy, fx] = 0; e[0, 0, ty, tx] = 1; add('food_under_the_tail', e)
    e = base.clone(); e[0, 2, free[4][0], free[4][1]] = L + 1; add('stray_larger_body_value', e)
    return out


def multi_variants(K, S, seed):
    torch.manual_seed(seed)
    env = MultiSnake(num_envs=1, num_snakes=K, size=S, device='cpu', manual_setup=False)
    f0, h0, b0, d0 = env.foods.clone(), env.heads.clone(), env.bodies.clone(), env.dones.clone()
    out = []

    def add(name, f, h, b, d):
        out.append((name, f.clone(), h.clone(), b.clone(), d.clone()))

    def cell(t, s, v):
        ys, xs = np.nonzero(t[s, 0].numpy() == v)
        return int(ys[0]), int(xs[0])

    occupied = (f0[0, 0] + h0[:, 0].sum(0) + b0[:, 0].sum(0)).numpy()
    free = [(y, x) for y in range(1, S - 1) for x in range(1, S - 1) if occupied[y, x] == 0]
    add('valid', f0, h0, b0, d0)
    b = b0.clone(); b[0, 0] = torch.max(b[0, 0], b0[1, 0]); add('snake_0_over_snake_1', f0, h0, b, d0)
    b = b0.clone(); h = h0.clone(); b[1] = b0[0]; h[1] = h0[0]; add('snake_1_on_top_of_snake_0', f0, h, b, d0)
    d = d0.clone(); d[1] = 1; add('dead_snake_with_a_body', f0, h0, b0, d)
    d = d0.clone(); d[1] = 1; h = h0.clone(); b = b0.clone(); h[1] = 0; b[1] = 0; add('dead_snake_cleared', f0, h, b, d)
    h = h0.clone(); h[0, 0, free[0][0], free[0][1]] = 1; add('two_heads_for_snake_0', f0, h, b0, d0)
    h = h0.clone(); h[2] = 0; add('no_head_for_snake_2', f0, h, b0, d0)
    y, x = cell(b0, 1, 2)
    b = b0.clone(); b[1, 0, y, x] = 0; add('body_value_2_missing_in_snake_1', f0, h0, b, d0)
    ty, tx = cell(b0, 0, 1)
    h = h0.clone(); h[0] = 0; h[0, 0, ty, tx] = 1; add('head_of_snake_0_on_its_tail', f0, h, b0, d0)
    hy, hx = cell(b0, 0, int(b0[0].max()))
    f = f0.clone(); f[0, 0, hy, hx] = 1; add('food_under_the_head_of_snake_0', f, h0, b0, d0)
    f = f0.clone(); f[0, 0, free[1][0], free[1][1]] = 2; add('food_value_2', f, h0, b0, d0)
    b = b0.clone(); b[2, 0] = 0; h = h0.clone(); h[2, 0] = 0; add('living_snake_2_without_cells', f0, h, b, d0)
    d = torch.ones_like(d0); add('all_dead_nothing_cleared', f0, h0, b0, d)
    d = torch.ones_like(d0); add('all_dead_all_cleared', f0, torch.zeros_like(h0), torch.zeros_like(b0), d)
    b = b0.clone(); b[0, 0, ty, tx] = 0; y2, x2 = cell(b0, 0, 2); b[0, 0, y2, x2] = 1; b[0, 0, hy, hx] = 2
    add('snake_0_of_length_2', f0, h0, b, d0)
    return out


def main():
    names, envs, snake_msg, env_msg = [], [], [], []
    for S, seed in ((9, 1), (12, 2)):
        for name, e in single_variants(S, seed):
            names.append(f'S{S}:{name}')
            envs.append((S, e.numpy()))
            snake_msg.append(verdict(lambda: ref_utils.snake_consistency(e)))
            env_msg.append(verdict(lambda: ref_utils.env_consistency(e)))
    out = {'single_names': np.array(names), 'single_snake_msg': np.array(snake_msg), 'single_env_msg': np.array(env_msg)}
    for i, (S, e) in enumerate(envs):
        out[f'single_env_{i}'] = e.astype(np.float32)
    # the reference on the whole batch of one size: the FIRST failing check in its own order decides the message
    for S in (9, 12):
        batch = torch.tensor(np.concatenate([e for s, e in envs if s == S]))
        out[f'single_batch_msg_S{S}'] = np.array(verdict(lambda: ref_utils.env_consistency(batch)))

    K, S = 3, 12
    mnames, mmsg = [], []
    for i, (name, f, h, b, d) in enumerate(multi_variants(K, S, 3)):
        env = MultiSnake(num_envs=1, num_snakes=K, size=S, device='cpu', manual_setup=True)
        env.foods, env.heads, env.bodies, env.dones = f, h, b, d
        mnames.append(name)
        mmsg.append(verdict(env.check_consistency))
        out[f'multi_foods_{i}'], out[f'multi_heads_{i}'] = f.numpy().astype(np.float32), h.numpy().astype(np.float32)
        out[f'multi_bodies_{i}'], out[f'multi_dones_{i}'] = b.numpy().astype(np.float32), d.numpy().astype(np.uint8)
    out['multi_names'], out['multi_msg'], out['multi_shape'] = np.array(mnames), np.array(mmsg), np.array([K, S])
    path = os.path.join(HERE, 'checker_failing_states.npz')
    np.savez_compressed(path, **out)
    print(f'{path}: {os.path.getsize(path) / 1024:.0f} KiB')
    for n, a, b in zip(names, snake_msg, env_msg):
        print(f'{n:40s} snake: {a[:60]!r:64s} env: {b[:60]!r}')
    for n, m in zip(mnames, mmsg):
        print(f'multi {n:36s} {m[:70]!r}')


if __name__ == '__main__':
    main()
