"""torch-op restatement of the reference's SingleSnake step / reset / partial_n observation — the CPU BASELINE leg.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): imported by bench.py's `cpu_baseline` leg and
by tests/test_torch_port.py.  The product package (wurm_amd/) never imports it.

Why it exists: the reference is Python and cannot travel to the GPU box,
so "the reference's own torch-CPU path timed on the host cores of the same box" is reported through a restatement
of the SAME KIND of program — batched torch ops over an (N,3,S,S) fp32 tensor: 3x3 convolutions for the head shift
and the orientation test, masks for decay / collisions / growth, `multinomial` for the food cell, two host syncs per
step — written from the transition spec of SURVEY.md Appendix A.1 / A.5 (which restates wurm/envs/single_snake.py:197-387
and wurm/utils.py:36-65), not from the reference's source.  tests/test_torch_port.py replays the random outcomes
this port draws into the scalar C oracle (pinned to the reference by the golden fixtures) and requires bit-equal
states, sanitised actions, rewards, dones and observations on every step, so it computes the reference's function.

Domain: well-formed snakes only (every step is followed by reset(done), as in the benchmark loop of
tests/test_single_snake_env.py:24-31); observation modes 'partial_n' and None.
"""
import torch
import torch.nn.functional as F

# head = neck + H[o] (orientation o); action a moves the head by -H[a] = [(+1,0),(0,-1),(-1,0),(0,+1)]
_H = [(-1, 0), (0, 1), (1, 0), (0, -1)]


def _shift_filters(offsets):
    """3x3 cross-correlation weights w with conv2d(x, w, padding=1)[y, x] = x[y - dy, x - dx] for each (dy, dx)."""
    w = torch.zeros(len(offsets), 1, 3, 3)
    for i, (dy, dx) in enumerate(offsets):
        w[i, 0, 1 - dy, 1 - dx] = 1.0
    return w


class TorchSingleSnake(object):
    def __init__(self, num_envs: int, size: int, observation_mode: str = 'partial_2', seed: int = 0):
        self.N, self.S = num_envs, size
        self.observation_mode = observation_mode
        self.gen = torch.Generator().manual_seed(seed)
        S = size
        self.orient_w = _shift_filters(_H)                          # neck shifted onto the head, per orientation
        self.move_w = _shift_filters([(-dy, -dx) for dy, dx in _H])  # head shifted by the action's step
        # a length-3 snake around a seed cell, per direction d: 3 at seed + H[d], 2 at the seed, 1 at seed - H[d]
        self.len3_w = 3 * _shift_filters(_H) + _shift_filters([(-dy, -dx) for dy, dx in _H])
        self.len3_w[:, 0, 1, 1] = 2.0
        self.interior = torch.zeros(S, S)
        self.interior[1:-1, 1:-1] = 1.0
        self.envs = torch.zeros(num_envs, 3, S, S)
        # what the last step / reset drew, for replay into the oracle: food cell (-1 = none); (sy, sx, d, food cell)
        self.last_food = torch.full((num_envs,), -1, dtype=torch.int32)
        self.last_reset = torch.full((num_envs, 4), -1, dtype=torch.int32)
        self.reset(torch.ones(num_envs, dtype=torch.bool), observe=False)

    # ------------------------------------------------------------------ helpers
    def _random_free_cell(self, envs):
        """One uniformly random interior cell with nothing on it per env of `envs` (n,3,S,S); -1 where none is free."""
        free = ((envs.sum(1) == 0) & (self.interior > 0)).flatten(1).float()
        has = free.sum(1) > 0
        cell = torch.full((envs.shape[0],), -1, dtype=torch.long)
        if bool(has.any()):
            cell[has] = torch.multinomial(free[has], 1, generator=self.gen).squeeze(1)
        return cell

    def _observe(self):
        if self.observation_mode is None:
            return None
        n = int(self.observation_mode.split('_')[-1])
        food, head, body = self.envs[:, 0], self.envs[:, 1], self.envs[:, 2]
        N, S, W = self.N, self.S, 2 * n + 1
        rgb = torch.ones(N, 3, S, S)
        occupied = (body > 0)[:, None]
        rgb = torch.where(occupied, torch.tensor([0.0, 127.0, 0.0]).div(255.0).view(1, 3, 1, 1), rgb)
        rgb = torch.where((head > 0)[:, None], torch.tensor([0.0, 1.0, 0.0]).view(1, 3, 1, 1), rgb)
        rgb = torch.where((food > 0)[:, None], torch.tensor([1.0, 0.0, 0.0]).view(1, 3, 1, 1), rgb)
        rgb = rgb * self.interior
        padded = F.pad(rgb, (n, n, n, n))
        flat = head.flatten(1)
        has_head = flat.sum(1) > 0
        hc = flat.argmax(1)
        hy, hx = hc // S, hc % S
        ar = torch.arange(W)
        rows = (hy[:, None] + ar)[:, None, :, None]
        cols = (hx[:, None] + ar)[:, None, None, :]
        crop = padded[torch.arange(N)[:, None, None, None], torch.arange(3)[None, :, None, None], rows, cols]
        crop = crop * has_head[:, None, None, None]
        return crop.reshape(N, 3 * W * W)

    # ------------------------------------------------------------------ step
    def step(self, actions: torch.Tensor):
        """actions (N,) int64, sanitised in place.  Returns obs, reward (N,1), done (N,1), info."""
        N = self.N
        food, head, body = self.envs[:, 0], self.envs[:, 1], self.envs[:, 2]
        L = body.flatten(1).amax(1)
        neck = (body == (L - 1)[:, None, None]).float()
        resp = (F.conv2d(neck[:, None], self.orient_w, padding=1) * head[:, None]).sum((2, 3))
        o = resp.argmax(1)
        actions.add_((o == actions).long() * 2).fmod_(4)
        newhead = F.conv2d(head[None], self.move_w[actions % 4], padding=1, groups=N)[0]
        eat = (newhead * food).flatten(1).sum(1) > 0
        body = torch.where(eat[:, None, None], body, F.relu(body - 1))
        selfc = (newhead * body).flatten(1).sum(1) > 0
        body = body + newhead * (L + eat.float())[:, None, None]
        food = food - newhead * food
        self.envs = torch.stack([food, newhead, body], 1)
        self.last_food.fill_(-1)
        if bool(eat.any()):
            cell = self._random_free_cell(self.envs[eat])
            idx = eat.nonzero().squeeze(1)
            ok = cell >= 0
            self.envs[:, 0].flatten(1)[idx[ok], cell[ok]] = 1.0
            self.last_food[idx] = cell.to(torch.int32)
        edge = (newhead * self.interior).flatten(1).sum(1) == 0
        done = selfc | edge
        info = {'self_collision': selfc, 'edge_collision': edge}
        return self._observe(), eat.float().unsqueeze(-1), done.unsqueeze(-1), info

    # ------------------------------------------------------------------ reset
    def reset(self, done: torch.Tensor, observe: bool = True):
        done = done.view(-1).bool()
        self.last_reset.fill_(-1)
        n = int(done.sum())
        if n > 0:
            S = self.S
            sy = torch.randint(4, S - 4, (n,), generator=self.gen) if S > 8 else None
            sx = torch.randint(4, S - 4, (n,), generator=self.gen)
            d = torch.randint(4, (n,), generator=self.gen)
            seed = torch.zeros(n, S * S)
            seed[torch.arange(n), sy * S + sx] = 1.0
            body = F.conv2d(seed.view(1, n, S, S), self.len3_w[d], padding=1, groups=n)[0]
            fresh = torch.stack([torch.zeros_like(body), (body == 3).float(), body], 1)
            cell = self._random_free_cell(fresh)
            fresh[:, 0].flatten(1)[torch.arange(n), cell] = 1.0
            self.envs[done] = fresh
            self.last_reset[done] = torch.stack([sy, sx, d, cell], 1).to(torch.int32)
        return self._observe() if observe else None
