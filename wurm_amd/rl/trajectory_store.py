"""TrajectoryStore — same interface as the reference's wurm.rl.TrajectoryStore (wurm/rl/trajectory_store.py:4-89),
with preallocated (capacity, ...) device buffers instead of Python lists + `torch.stack` on every read: `append`
copies one (num_envs, ...) row into the next slot, the properties return the filled prefix as a view.

Rows that carry autograd history (log_prob, value, entropy) keep it: the copy into the buffer is an in-place op that
autograd tracks; `clear()` drops the buffers' history so the next rollout starts a fresh graph.

Aliasing: a tensor read from a property is a view of the buffer.  Like the reference's `torch.stack` copies it must stay
valid after `clear()` and the next rollout (callers keep trajectories for logging or a delayed backward), so `clear()`
lets go of every buffer that has been handed out since the last clear and the next `append` allocates a new one; buffers
nobody looked at are reused."""
import torch

_FIELDS = ('state', 'action', 'log_prob', 'reward', 'value', 'done', 'entropy', 'hidden_state')


class TrajectoryStore(object):
    """Stores transitions; each property returns a tensor of shape (num_steps, ...) (reference :4-8)."""

    def __init__(self, capacity: int = 64):
        self.capacity = capacity
        self._buf = {}
        self._len = {}
        self._handed_out = set()
        self.clear()

    def _put(self, name: str, x: torch.Tensor):
        n = self._len.get(name, 0)
        buf = self._buf.get(name)
        if buf is None or buf.shape[1:] != x.shape or buf.dtype != x.dtype or buf.device != x.device:
            buf = torch.empty((self.capacity,) + tuple(x.shape), dtype=x.dtype, device=x.device)
            n = 0
        elif n == buf.shape[0]:  # full: grow geometrically
            buf = torch.cat([buf, torch.empty_like(buf)], dim=0)
        elif n == 0 and buf.grad_fn is not None:
            buf = buf.detach()
        buf[n].copy_(x)
        self._buf[name] = buf
        self._len[name] = n + 1

    def append(self, state=None, action=None, log_prob=None, reward=None, value=None, done=None, entropy=None,
               hidden_state=None):
        """Adds a transition to the store; each argument is a (num_envs, ...) tensor (reference :12-48)."""
        for name, x in zip(_FIELDS, (state, action, log_prob, reward, value, done, entropy, hidden_state)):
            if x is not None:
                self._put(name, x)

    def clear(self):
        """reference :50-58"""
        self._buf = {k: v.detach() for k, v in self._buf.items() if k not in self._handed_out}
        self._len = {k: 0 for k in self._buf}
        self._handed_out = set()

    def _get(self, name: str) -> torch.Tensor:
        n = self._len.get(name, 0)
        if n == 0:
            raise RuntimeError('stack expects a non-empty TensorList')  # what torch.stack([]) raises in the reference
        self._handed_out.add(name)
        return self._buf[name][:n]

    states = property(lambda self: self._get('state'))
    actions = property(lambda self: self._get('action'))
    log_probs = property(lambda self: self._get('log_prob'))
    rewards = property(lambda self: self._get('reward'))
    values = property(lambda self: self._get('value'))
    dones = property(lambda self: self._get('done'))
    entropies = property(lambda self: self._get('entropy'))
    hidden_state = property(lambda self: self._get('hidden_state'))
