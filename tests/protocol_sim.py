"""A SIMULATING stand-in for libwurm_hip.so, for testing the host protocols of the env classes without a GPU.

The recording stand-ins of tests/test_host_lazy_reset.py / test_host_multi_mirror.py check the SEQUENCE of entry points a
class asks for; this one plays the library's side on real (CPU) memory with a hash algebra in place of the dynamics, so
that two env objects driven by the same caller events can be compared on the VALUES the caller sees:

  * the state of env e is a 63-bit token h[e] (low bit: "fresh", i.e. just rebuilt); a state tensor row holds an
    expansion of the token (so a caller's in-place edit of any cell makes it another state), the compact mirror holds the
    token itself;
  * step:  h' = H('step', h, call, action, global env id [, cfg]);  done / reward / info = functions of h';
           observation = expansion of H('obs', h', mode, n);
    reset: h' = H('reset', h, call, ...) | fresh for the envs flagged, unchanged for the others;
    rollout(T) = T x (step; reset(done)) with counters call0 + 2t, call0 + 2t + 1 — the identities the real kernels
    guarantee bit-for-bit (include/wurm_hip.h), and nothing else;
  * the mirror protocol of wurm_single_call / wurm_multi_call is followed to the letter: a launch given `resident` with
    `resident_valid` reads ONLY the mirror, a lazy launch does not write the tensors, the flush entry points write them.

A deferred reset applied with the wrong counter, mask, start location or configuration, an observation pre-computed in
another mode, a mirror used after a foreign write, a lazy mirror not written out before something reads the tensors: each
of them changes a token, hence some value the caller sees (tests/test_protocol_enumeration.py compares against a twin object
with `lazy_reset=False, resident_mirror=False`).  Test infrastructure only.
"""
import ctypes
import hashlib

import numpy as np

from wurm_amd import _lib

OBS_DEFAULT, OBS_RAW, OBS_ONE_CHANNEL, OBS_POSITIONS, OBS_PARTIAL, OBS_NONE = range(6)
MASK64 = (1 << 64) - 1


def _addr(p) -> int:
    if p is None:
        return 0
    v = getattr(p, 'value', p)
    return int(v or 0)


def _int(x) -> int:
    return int(getattr(x, 'value', x))


def mem(p, count, dtype=np.uint8):
    """writable numpy view of `count` items of `dtype` at address p"""
    nbytes = int(count) * np.dtype(dtype).itemsize
    buf = (ctypes.c_char * nbytes).from_address(_addr(p))
    return np.frombuffer(buf, dtype=dtype)


def H(*parts) -> int:
    """62-bit hash of the parts (ints / bytes / strings).  A state TOKEN is such a hash shifted left by two, bit 1 = "the
    checker finds fault with this state", bit 0 = "fresh" (just rebuilt by a reset; always consistent)."""
    h = hashlib.blake2b(digest_size=8)
    for p in parts:
        if isinstance(p, (bytes, bytearray)):
            h.update(b'b' + len(p).to_bytes(4, 'little') + bytes(p))
        elif isinstance(p, str):
            h.update(b's' + p.encode() + b'\0')
        else:
            h.update(b'i' + (int(p) & MASK64).to_bytes(8, 'little'))
    return int.from_bytes(h.digest(), 'little') >> 2


_expand_cache = {}


def expand(tok: int, n: int) -> np.ndarray:
    """n floats, each a small integer: the token's 8 bytes, then a stream derived from it"""
    key = (tok, n)
    out = _expand_cache.get(key)
    if out is None:
        raw = (tok & MASK64).to_bytes(8, 'little') + hashlib.shake_128(b'x' + (tok & MASK64).to_bytes(8, 'little')).digest(max(n - 8, 0))
        out = np.frombuffer(raw[:n], dtype=np.uint8).astype(np.float32)
        if len(_expand_cache) > 200000:
            _expand_cache.clear()
        _expand_cache[key] = out
    return out


def token_of(row: np.ndarray) -> int:
    """the token a row of floats is the expansion of; anything else (a caller's edit) is a state of its own, not fresh"""
    n = row.shape[0]
    head = row[:8]
    if n >= 8 and np.all(head >= 0) and np.all(head < 256) and np.all(head == np.floor(head)):
        tok = int.from_bytes(head.astype(np.uint8).tobytes(), 'little')
        if np.array_equal(row, expand(tok, n)):
            return tok
    return tok_of_hash(H('edited', row.tobytes()))


def tok_of_hash(hv: int) -> int:
    """a non-fresh state: one in 16 of them is one the checker rejects"""
    return ((hv << 2) | (2 if hv % 16 == 0 else 0)) & MASK64


def step_token(h, call, action, gid, extra=b''):
    return tok_of_hash(H('step', h, call, action, gid, extra))


def reset_token(h, call, gid, extra=b''):
    return ((H('reset', h, call, gid, extra) << 2) | 1) & MASK64


def bit(h, what) -> int:
    return H(what, h) & 1


def done_bit(h) -> int:
    return int(H('done', h) % 3 == 0)     # a third of the steps end an episode: resets matter in short sequences


def check_err(h) -> int:
    """the checker's verdict on a state (bit 5: 'A snake has size of less than 3.')"""
    return 0x20 if (h & 2) else 0


def single_obs_elems(m, n, S, C=3):
    if m == OBS_DEFAULT:
        return 3 * S * S
    if m == OBS_RAW:
        return C * S * S
    if m == OBS_ONE_CHANNEL:
        return S * S
    if m == OBS_POSITIONS:
        return 4
    if m == OBS_PARTIAL:
        return 3 * (2 * n + 1) ** 2
    return 0


class SimSingle(object):
    """libwurm_hip.so's wurm_single_* / wurm_grid_* entry points over the hash algebra.  `channels` 3 = SingleSnake,
    2 = SimpleGridworld (no mirror, no in-place action sanitising, start location part of every reset)."""

    def __init__(self, channels=3, mirror_auto=True):
        self.C = channels
        self.mirror_auto = mirror_auto       # wurm_single_resident_bytes offers the mirror (batch "above the threshold")
        self.calls = []
        self.fail_next = False

    # ---- state access
    def _rows(self, envs, N, S):
        return mem(envs, N * self.C * S * S, np.float32).reshape(N, self.C * S * S)

    def read_state(self, envs, N, S):
        rows = self._rows(envs, N, S)
        return [token_of(rows[e]) for e in range(N)]

    def write_state(self, envs, N, S, toks):
        rows = self._rows(envs, N, S)
        for e in range(N):
            rows[e] = expand(toks[e], rows.shape[1])

    def _write_obs(self, obs, toks, m, n, S):
        if not _addr(obs) or m == OBS_NONE:
            return
        E = single_obs_elems(m, n, S, self.C)
        o = mem(obs, len(toks) * E, np.float32).reshape(len(toks), E)
        for e, h in enumerate(toks):
            o[e] = expand(H('obs', h, m, n), E) if E >= 8 else expand(H('obs', h, m, n), 8)[:E]

    # ---- entry points
    def _reset(self, name, envs, done, obs, m, n, N, S, call, off, start):
        N, S, call, off, m, n = _int(N), _int(S), _int(call), _int(off), _int(m), _int(n)
        self.calls.append(name)
        toks = self.read_state(envs, N, S)
        d = mem(done, N)
        toks = [reset_token(h, call, off + e, start) if d[e] else h for e, h in enumerate(toks)]
        self.write_state(envs, N, S, toks)
        self._write_obs(obs, toks, m, n, S)
        return 0

    def wurm_single_reset(self, envs, done, obs, m, n, N, S, seed, call, off, inj, stream):
        return self._reset('reset', envs, done, obs, m, n, N, S, call, off, b'')

    def wurm_grid_reset(self, envs, done, obs, m, n, N, S, sy, sx, seed, call, off, inj, stream):
        return self._reset('reset', envs, done, obs, m, n, N, S, call, off, b'%d,%d' % (_int(sy), _int(sx)))

    def _observe(self, envs, obs, m, n, N, S, stream):
        N, S = _int(N), _int(S)
        self.calls.append('observe')
        self._write_obs(obs, self.read_state(envs, N, S), _int(m), _int(n), S)
        return 0

    wurm_single_observe = _observe
    wurm_grid_observe = _observe

    def wurm_single_check(self, envs, err, N, S, stream):
        N, S = _int(N), _int(S)
        self.calls.append('check')
        out = mem(err, N, np.uint32)
        for e, h in enumerate(self.read_state(envs, N, S)):
            out[e] = check_err(h)
        return 0

    def wurm_single_resident_bytes(self, N, S, m, n):
        return 8 * _int(N) if (self.mirror_auto and self.C == 3) else 0

    def wurm_single_resident_size(self, N, S, m, n):
        return 8 * _int(N) if self.C == 3 else 0

    def wurm_single_resident_flush(self, c_addr, stream):
        c = _lib.SingleCall.from_address(_addr(c_addr))
        if c.resident and c.resident_lazy and c.resident_valid == 1:
            self.calls.append('flush')
            N = c.num_envs
            self.write_state(c.envs, N, c.size, [int(x) for x in mem(c.resident, N, np.uint64)])
        return 0

    # SimpleGridworld's mirror (round 6): same protocol, plus resident_valid == 2 — REFUSED: the launch that built the mirror
    # found envs it cannot describe, the planes stay the state (`refuse_builds`: every build of this simulator ends that way)
    refuse_builds = False

    def wurm_grid_resident_bytes(self, N, S, m):
        return 8 * _int(N) if (self.mirror_auto and self.C == 2) else 0

    def wurm_grid_resident_size(self, N, S, m):
        return 8 * _int(N) if self.C == 2 else 0

    wurm_grid_resident_flush = wurm_single_resident_flush

    def _actions(self, actions, dtype, count):
        return mem(actions, count, np.int64 if dtype == 0 else np.int32)

    def _one_step(self, h, call, a, gid):
        """(new token, sanitised action, reward, done, self_collision, edge_collision)"""
        if self.C == 3:
            a = (a + 2 * bit(H('san', h, a), 'flip')) % 4      # "reverse moves become forward moves", in place
        h2 = step_token(h, call, a, gid)
        return h2, a, float(H('rew', h2) % 2), done_bit(h2), bit(h2, 'selfc'), bit(h2, 'edgec')

    def step_slot(self, c_addr, sl_addr, slot, actions, dtype, call, pending, pre_call, want_after, stream):
        c = _lib.SingleCall.from_address(_addr(c_addr))
        sl = _lib.SingleSlabs.from_address(_addr(sl_addr))
        N, S, off = c.num_envs, c.size, c.env_offset
        slot, call, pre_call, dtype = _int(slot), _int(call), _int(pre_call), _int(dtype)
        if self.fail_next:
            self.fail_next = False
            if c.resident:
                c.resident_valid = 0
            return -3
        assert 0 <= slot < sl.steps
        start = b'%d,%d' % (c.start_y, c.start_x) if self.C == 2 else b''
        from_mirror = bool(c.resident) and c.resident_valid == 1
        refused = bool(c.resident) and (c.resident_valid == 2 or (self.refuse_builds and not from_mirror))
        self.calls.append('step')
        toks = [int(x) for x in mem(c.resident, N, np.uint64)] if from_mirror else self.read_state(c.envs, N, S)
        if pending:
            d = mem(c.done_copy, N)
            toks = [reset_token(h, pre_call, off + e, start) if d[e] else h for e, h in enumerate(toks)]
        act = self._actions(actions, dtype, N)
        E = single_obs_elems(c.obs_mode, c.obs_n, S, self.C)
        reward = mem(sl.reward + 4 * slot * N, N, np.float32)
        flags = mem(sl.flags, 3 * sl.steps * N).reshape(3, sl.steps, N)
        new, dn = [], []
        for e, h in enumerate(toks):
            h2, a, r, d, sc, ec = self._one_step(h, call, int(act[e]), off + e)
            act[e] = a
            reward[e] = r
            flags[0, slot, e], flags[2, slot, e] = d, ec
            if self.C == 3:
                flags[1, slot, e] = sc
            new.append(h2)
            dn.append(d)
        if c.done_copy:
            mem(c.done_copy, N)[:] = dn
        self._write_obs(sl.obs + 4 * slot * N * E, new, c.obs_mode, c.obs_n, S)
        if want_after:
            assert sl.obs_after, 'want_obs_after without a slab for it'
            after = [reset_token(h, call + 1, off + e, start) if dn[e] else h for e, h in enumerate(new)]
            self._write_obs(sl.obs_after + 4 * slot * N * E, after, c.obs_mode, c.obs_n, S)
        if c.check_mask:
            cm = mem(c.check_mask, N, np.uint32)
            for e, h in enumerate(new):
                # only the mirror-resident kernel computes them; a finished env is never vouched for
                cm[e] = 0xFFFFFFFF if (dn[e] or not c.resident) else check_err(h)
        if c.resident and not refused:
            mem(c.resident, N, np.uint64)[:] = np.array(new, dtype=np.uint64)
            c.resident_valid = 1
            if not c.resident_lazy or (self.C == 2 and not from_mirror):   # (SimpleGridworld: a launch that BUILDS the mirror writes the planes whatever `lazy` says)
                self.write_state(c.envs, N, S, new)
        else:
            if refused:
                c.resident_valid = 2
            self.write_state(c.envs, N, S, new)
        return 0

    def _rollout(self, envs, actions, dtype, reward, done, selfc, edgec, obs, m, n, N, S, T, call0, off, start):
        N, S, T, call0, off, m, n, dtype = (_int(x) for x in (N, S, T, call0, off, m, n, dtype))
        self.calls.append('rollout')
        toks = self.read_state(envs, N, S)
        act = self._actions(actions, dtype, T * N).reshape(T, N) if T else None
        E = single_obs_elems(m, n, S, self.C)
        for t in range(T):
            new, dn = [], []
            for e, h in enumerate(toks):
                h2, a, r, d, sc, ec = self._one_step(h, call0 + 2 * t, int(act[t, e]), off + e)
                act[t, e] = a
                mem(reward + 4 * (t * N + e), 1, np.float32)[0] = r
                mem(done + t * N + e, 1)[0] = d
                mem(edgec + t * N + e, 1)[0] = ec
                if selfc is not None:
                    mem(selfc + t * N + e, 1)[0] = sc
                new.append(h2)
                dn.append(d)
            if _addr(obs):
                self._write_obs(_addr(obs) + 4 * t * N * E, new, m, n, S)
            toks = [reset_token(h, call0 + 2 * t + 1, off + e, start) if dn[e] else h for e, h in enumerate(new)]
        self.write_state(envs, N, S, toks)
        return 0

    def wurm_single_rollout(self, envs, actions, dtype, reward, done, selfc, edgec, obs, m, n, N, S, T, seed, call0, off,
                            inj_f, inj_r, stream):
        return self._rollout(_addr(envs), actions, dtype, _addr(reward), _addr(done), _addr(selfc), _addr(edgec), obs, m, n,
                             N, S, T, call0, off, b'')

    def wurm_grid_rollout(self, envs, actions, dtype, reward, done, edgec, obs, m, n, N, S, T, sy, sx, seed, call0, off,
                          inj_f, inj_r, stream):
        return self._rollout(_addr(envs), actions, dtype, _addr(reward), _addr(done), None, _addr(edgec), obs, m, n,
                             N, S, T, call0, off, b'%d,%d' % (_int(sy), _int(sx)))

    rollout_serves_mirror = True   # False: a batch / observation the lane rollout does not serve (the library's fallback)

    def wurm_grid_rollout_resident(self, envs, actions, dtype, reward, done, edgec, obs, m, n, N, S, T, sy, sx, seed, call0, off,
                                   resident, valid_addr, lazy, stream):
        """round 6: the rollout on SimpleGridworld's mirror — the state from the records when they are current, the records
        current afterwards, the planes written unless (lazy and the records were current); a launch that builds the mirror may
        refuse it (2); the fallback writes a lazy valid mirror out first and leaves the mirror stale"""
        start = b'%d,%d' % (_int(sy), _int(sx))
        args = (actions, dtype, _addr(reward), _addr(done), None, _addr(edgec), obs, m, n, N, S, T, call0, off, start)
        return self._rollout_resident(envs, args, N, S, T, resident, valid_addr, lazy)

    def wurm_single_rollout_resident(self, envs, actions, dtype, reward, done, selfc, edgec, obs, m, n, N, S, T, seed, call0, off,
                                     resident, valid_addr, lazy, stream):
        """... and SingleSnake's (grids of 12 x 12 and larger keep the mirror: rollout_serves_mirror; 9 x 9: the fallback)"""
        args = (actions, dtype, _addr(reward), _addr(done), _addr(selfc), _addr(edgec), obs, m, n, N, S, T, call0, off, b'')
        return self._rollout_resident(envs, args, N, S, T, resident, valid_addr, lazy)

    def _rollout_resident(self, envs, args, N, S, T, resident, valid_addr, lazy):
        Ni, Si, lazy = _int(N), _int(S), _int(lazy)
        valid = ctypes.c_int.from_address(_addr(valid_addr))
        mir = mem(resident, Ni, np.uint64)
        if valid.value != 2 and self.rollout_serves_mirror and _int(T) > 0:
            from_mirror = valid.value == 1
            self.calls.append('rollout_resident')
            planes = mem(_addr(envs), Ni * self.C * Si * Si * 4)
            held = None
            if from_mirror:
                # (the stand-in rolls out in the planes' memory: the records' state is brought there — and, lazy, what the
                # planes held is put back afterwards: the library does not write them)
                if lazy:
                    held = bytes(planes)
                self.write_state(_addr(envs), Ni, Si, [int(x) for x in mir])
            rc = self._rollout(_addr(envs), *args)
            if self.refuse_builds and not from_mirror:
                valid.value = 2          # (a build that met envs it cannot describe: the planes are the state)
            else:
                mir[:] = np.array(self.read_state(_addr(envs), Ni, Si), dtype=np.uint64)
                valid.value = 1
                if held is not None:
                    planes[:] = np.frombuffer(held, dtype=np.uint8)
            return rc
        if lazy and valid.value == 1:
            self.calls.append('flush')
            self.write_state(_addr(envs), Ni, Si, [int(x) for x in mir])
        if valid.value != 2:
            valid.value = 0
        return self._rollout(_addr(envs), *args)


_STEP_SLOT_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p)
_MULTI_SLOT_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                 ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p)


class CMultiSlot(object):
    """a stand-in's wurm_multi_step_slot behind a real C function pointer (wurm_amd._fastcall.Stepper, multi side)"""

    def __init__(self, sim):
        self._cb = _MULTI_SLOT_T(lambda c, sl, slot, a, call, pend, pre, want, st: sim.step_slot(c, sl, slot, a, call, pend, pre,
                                                                                         want, st))
        self.c_address = ctypes.cast(self._cb, ctypes.c_void_p).value

    def __call__(self, *args):
        return self._cb(*args)


class CSlot(object):
    """a stand-in's step_slot behind a real C function pointer, as wurm_amd._fastcall.Stepper needs it"""

    def __init__(self, sim):
        self._cb = _STEP_SLOT_T(lambda c, sl, slot, a, dt, call, pend, pre, want, st: sim.step_slot(
            c, sl, slot, a, dt, call, pend, pre, want, st))
        self.c_address = ctypes.cast(self._cb, ctypes.c_void_p).value

    def __call__(self, *args):
        return self._cb(*args)


# ------------------------------------------------------------------------------------------------ MultiSnake

def multi_obs_elems(m, n, S):
    return 3 * S * S if m == OBS_DEFAULT else (3 * (2 * n + 1) ** 2 if m == OBS_PARTIAL else 0)


def _tb(tok):
    return (tok & MASK64).to_bytes(8, 'little')


class SimMulti(object):
    """wurm_multi_* over the hash algebra.  The state of env e is (M, R): M the token of the MIRRORED tensors (foods row e,
    heads / bodies rows e*K..e*K+K-1), R whatever bytes dones / orientations / colours hold for the env (the kernels always
    read and write those in place; every launch that changes M rewrites R as a function of the new M)."""

    def __init__(self, mirror_auto=True):
        self.mirror_auto = mirror_auto
        self.calls = []
        self.rollout_keeps_mirror = True

    # ---- state access
    @staticmethod
    def _views(foods, heads, bodies, N, K, S):
        return (mem(foods, N * S * S, np.float32).reshape(N, S * S), mem(heads, N * K * S * S, np.float32).reshape(N, K * S * S),
                mem(bodies, N * K * S * S, np.float32).reshape(N, K * S * S))

    def read_m(self, foods, heads, bodies, N, K, S):
        f, hd, b = self._views(foods, heads, bodies, N, K, S)
        out = []
        for e in range(N):
            tok = token_of(f[e])
            if not (np.array_equal(hd[e], expand(H('heads', tok), hd.shape[1])) and
                    np.array_equal(b[e], expand(H('bodies', tok), b.shape[1]))):
                tok = tok_of_hash(H('edited', f[e].tobytes(), hd[e].tobytes(), b[e].tobytes()))
            out.append(tok)
        return out

    def write_m(self, foods, heads, bodies, N, K, S, toks):
        f, hd, b = self._views(foods, heads, bodies, N, K, S)
        for e, tok in enumerate(toks):
            f[e] = expand(tok, f.shape[1])
            hd[e] = expand(H('heads', tok), hd.shape[1])
            b[e] = expand(H('bodies', tok), b.shape[1])

    @staticmethod
    def _rest(dones, orientations, colours, N, K):
        return mem(dones, N * K).reshape(N, K), mem(orientations, N * K, np.int64).reshape(N, K), \
            mem(colours, N * K * 3, np.int16).reshape(N, K * 3)

    @staticmethod
    def _rest_tokens(rows, N):
        d, o, c = rows
        return [H('rest', d[e].tobytes(), o[e].tobytes(), c[e].tobytes()) for e in range(N)]

    @staticmethod
    def _write_rest(rows, toks, K, recolour):
        d, o, c = rows
        for e, tok in enumerate(toks):
            s = hashlib.shake_128(b'r' + _tb(tok)).digest(8 * K)
            d[e] = np.frombuffer(s[:K], np.uint8) & 1
            o[e] = np.frombuffer(s[K:2 * K], np.uint8) & 3
            if recolour:
                c[e] = np.frombuffer(s[2 * K:8 * K], np.int16)[:3 * K] & 0xff

    @staticmethod
    def _view_tokens(mt, rows, boost_rows, m):
        """what an observation of the state shows: the grids, who is dead, the colours and (crops) who boosted"""
        d, _, c = rows
        return [H('view', mt[e], d[e].tobytes(), c[e].tobytes(), boost_rows[e].tobytes() if m == OBS_PARTIAL else b'')
                for e in range(len(mt))]

    def _write_obs(self, obs, toks, m, n, N, K, S):
        if not _addr(obs) or m == OBS_NONE:
            return
        E = multi_obs_elems(m, n, S)
        o = mem(obs, K * N * E, np.float32).reshape(K, N, E)
        for e, h in enumerate(toks):
            for k in range(K):
                o[k, e] = expand(H('obs', h, k, m, n), E)

    @staticmethod
    def _cfg_bytes(cfg):
        c = getattr(cfg, '_obj', cfg)
        return bytes(c) if isinstance(c, ctypes.Structure) else bytes(_lib.MultiConfig.from_address(_addr(c)))

    @staticmethod
    def _reset_cfg(cfgb):
        """what wurm_multi_reset reads of the configuration: respawn_any and colour_random (multi_snake.hip: reroll_colour,
        multi_reset_grid; reference :800-831) — a change of the step dynamics between step and reset does not matter to it"""
        c = _lib.MultiConfig.from_buffer_copy(cfgb)
        return b'%d,%d' % (c.respawn_any, c.colour_random)

    @staticmethod
    def _reset_toks(mt, rt, d, call, off, cfgb):
        """after wurm_multi_reset(done_env = d): EVERY env passes through it (colours of snakes that are still dead are
        re-rolled, respawn 'any' acts on envs that are not flagged), so every token changes; flagged envs become fresh, the
        others keep the checker's verdict (a respawned snake is a well-formed one)"""
        out = []
        cfgb = SimMulti._reset_cfg(cfgb)
        for e, (m, r) in enumerate(zip(mt, rt)):
            hv = H('mreset', m, r, call, off + e, int(d[e]), cfgb)
            out.append(((hv << 2) | (1 if d[e] else (m & 2))) & MASK64)
        return out

    # ---- entry points
    def wurm_multi_colours(self, colours, N, K, fixed, seed, call, off, stream):
        N, K = _int(N), _int(K)
        c = mem(colours, N * K * 3, np.int16)
        c[:] = np.frombuffer(hashlib.shake_128(b'col%d,%d' % (_int(call), _int(fixed))).digest(2 * N * K * 3), np.int16) & 0xff
        return 0

    def wurm_multi_reset(self, foods, heads, bodies, dones, orientations, colours, done_env, status, boost, obs, m, n, N, K, S,
                         cfg, seed, call, off, inj, stream):
        N, K, S, m, n, call, off = (_int(x) for x in (N, K, S, m, n, call, off))
        self.calls.append('reset')
        d = mem(done_env, N)
        rows = self._rest(dones, orientations, colours, N, K)
        toks = self._reset_toks(self.read_m(foods, heads, bodies, N, K, S), self._rest_tokens(rows, N), d, call, off,
                                self._cfg_bytes(cfg))
        self.write_m(foods, heads, bodies, N, K, S, toks)
        self._write_rest(rows, toks, K, True)
        if _addr(obs) and m != OBS_NONE:
            b = mem(boost, N * K).reshape(N, K) if _addr(boost) else np.zeros((N, K), np.uint8)
            self._write_obs(obs, self._view_tokens(toks, rows, b, m), m, n, N, K, S)
        return 0

    def wurm_multi_observe(self, foods, heads, bodies, dones, boost, colours, obs, m, n, N, K, S, stream):
        N, K, S, m, n = (_int(x) for x in (N, K, S, m, n))
        self.calls.append('observe')
        rows = (mem(dones, N * K).reshape(N, K), None, mem(colours, N * K * 3, np.int16).reshape(N, K * 3))
        b = mem(boost, N * K).reshape(N, K)
        self._write_obs(obs, self._view_tokens(self.read_m(foods, heads, bodies, N, K, S), rows, b, m), m, n, N, K, S)
        return 0

    def wurm_multi_check(self, foods, heads, bodies, dones, err, N, K, S, stream):
        N, K, S = (_int(x) for x in (N, K, S))
        self.calls.append('check')
        mem(err, N, np.uint32)[:] = [check_err(h) for h in self.read_m(foods, heads, bodies, N, K, S)]
        return 0

    def wurm_multi_resident_bytes(self, N, K, S):
        return 8 * _int(N) if self.mirror_auto else 0

    def wurm_multi_resident_size(self, N, K, S):
        return 8 * _int(N)

    def wurm_multi_resident_flush(self, c_addr, stream):
        c = _lib.MultiCall.from_address(_addr(c_addr))
        if c.resident and c.resident_lazy and c.resident_valid:
            self.calls.append('flush')
            self.write_m(c.foods, c.heads, c.bodies, c.num_envs, c.num_snakes, c.size,
                         [int(x) for x in mem(c.resident, c.num_envs, np.uint64)])
        return 0

    @staticmethod
    def _step_toks(mt, rt, call, act_cols, off, cfgb):
        new = [tok_of_hash(H('mstep', mt[e], rt[e], call, act_cols[e], off + e, cfgb)) for e in range(len(mt))]
        return new, [done_bit(h) for h in new]

    @staticmethod
    def _packed(toks, K, N):
        """out_f32 (6K,N) floats and out_u8 (7K,N) flags of wurm_multi_step_packed as functions of the new tokens"""
        raw = hashlib.shake_128(b'o' + b''.join(_tb(t) for t in toks)).digest(13 * K * N)
        return (np.frombuffer(raw[:6 * K * N], np.uint8).astype(np.float32).reshape(6 * K, N),
                (np.frombuffer(raw[6 * K * N:], np.uint8) & 1).reshape(7 * K, N))

    def step_slot(self, c_addr, sl_addr, slot, actions, call, pending, pre_call, want_after, stream):
        """wurm_multi_step_slot: wurm_multi_step_packed on slot `slot` of the slabs"""
        c = _lib.MultiCall.from_address(_addr(c_addr))
        sl = _lib.MultiSlabs.from_address(_addr(sl_addr))
        slot = _int(slot)
        assert 0 <= slot < sl.steps
        KN = c.num_snakes * c.num_envs
        per_obs = 4 * KN * sl.obs_elems
        if want_after:
            assert sl.obs_after, 'want_obs_after without a slab for it'
        return self.step_packed(c_addr, sl.out_f32 + 4 * slot * 6 * KN, sl.out_u8 + slot * (7 * KN + c.num_envs),
                                sl.obs + slot * per_obs, sl.obs_after + slot * per_obs if want_after else None, actions, call,
                                pending, pre_call, stream)

    def step_packed(self, c_addr, of, ob, obs, obs_after, actions, call, pending, pre_call, stream):
        c = _lib.MultiCall.from_address(_addr(c_addr))
        N, K, S, off = c.num_envs, c.num_snakes, c.size, c.env_offset
        call, pre_call = _int(call), _int(pre_call)
        cfgb = bytes(c.cfg)
        self.calls.append('step')
        from_mirror = bool(c.resident) and bool(c.resident_valid)
        mt = [int(x) for x in mem(c.resident, N, np.uint64)] if from_mirror else self.read_m(c.foods, c.heads, c.bodies, N, K, S)
        rows = self._rest(c.dones, c.orientations, c.colours, N, K)
        rebuilt = [0] * N
        if pending:
            d = mem(c.all_done_copy, N)
            mt = self._reset_toks(mt, self._rest_tokens(rows, N), d, pre_call, off, cfgb)
            self._write_rest(rows, mt, K, True)
            rebuilt = [int(x) for x in d]
        act = mem(actions, K * N, np.int64).reshape(K, N)
        new, dn = self._step_toks(mt, self._rest_tokens(rows, N), call, [act[:, e].tobytes() for e in range(N)], off, cfgb)
        self._write_rest(rows, new, K, False)
        pf, pb = self._packed(new, K, N)
        mem(of, 6 * K * N, np.float32)[:] = pf.reshape(-1)
        b = mem(ob, 7 * K * N + N)
        b[:7 * K * N] = pb.reshape(-1)
        b[7 * K * N:] = dn
        boost_rows = b[:K * N].reshape(N, K)        # boost_this_step, env-major: the first block of out_u8
        if c.all_done_copy:
            mem(c.all_done_copy, N)[:] = dn
        self._write_obs(obs, self._view_tokens(new, rows, boost_rows, c.obs_mode), c.obs_mode, c.obs_n, N, K, S)
        if _addr(obs_after):
            # what wurm_multi_reset(all_done, call + 1) would return, on a COPY: that reset is not applied to the tensors
            rows2 = tuple(x.copy() for x in rows)
            after = self._reset_toks(new, self._rest_tokens(rows2, N), dn, call + 1, off, cfgb)
            self._write_rest(rows2, after, K, True)
            self._write_obs(obs_after, self._view_tokens(after, rows2, boost_rows, c.obs_mode), c.obs_mode, c.obs_n, N, K, S)
            if c.check_mask_after:
                mem(c.check_mask_after, N, np.uint32)[:] = [check_err(h) for h in after]
        if c.check_mask:
            # vouched for only where the image came from the mirror or from a rebuild
            mem(c.check_mask, N, np.uint32)[:] = [check_err(h) if (from_mirror or rebuilt[e]) else 0xFFFFFFFF
                                                  for e, h in enumerate(new)]
        if c.resident:
            mem(c.resident, N, np.uint64)[:] = np.array(new, dtype=np.uint64)
            c.resident_valid = 1
            if not c.resident_lazy:
                self.write_m(c.foods, c.heads, c.bodies, N, K, S, new)
        else:
            self.write_m(c.foods, c.heads, c.bodies, N, K, S, new)
        return 0

    def _rollout(self, dones, orientations, colours, boost, actions, out_f, out_b, all_done, obs, m, n, N, K, S, T, cfg, call0,
                 off, mt):
        cfgb = self._cfg_bytes(cfg)
        act = mem(actions, T * K * N, np.int64).reshape(T, K, N) if T else None
        E = multi_obs_elems(m, n, S)
        rows = self._rest(dones, orientations, colours, N, K)
        for t in range(T):
            new, dn = self._step_toks(mt, self._rest_tokens(rows, N), call0 + 2 * t, [act[t, :, e].tobytes() for e in range(N)],
                                      off, cfgb)
            self._write_rest(rows, new, K, False)
            pf, pb = self._packed(new, K, N)
            # the rollout's (T,3,K,N) / (T,4,K,N) blocks are the agent-major rows of the per-call packed layout
            mem(_addr(out_f) + 4 * t * 3 * K * N, 3 * K * N, np.float32)[:] = pf[3 * K:].reshape(-1)
            mem(_addr(out_b) + t * 4 * K * N, 4 * K * N)[:] = pb[3 * K:].reshape(-1)
            brow = mem(_addr(boost), N * K)
            brow[:] = pb[:K].reshape(-1)
            mem(_addr(all_done) + t * N, N)[:] = dn
            if _addr(obs) and m != OBS_NONE:
                self._write_obs(_addr(obs) + 4 * t * K * N * E, self._view_tokens(new, rows, brow.reshape(N, K), m), m, n, N, K, S)
            mt = self._reset_toks(new, self._rest_tokens(rows, N), dn, call0 + 2 * t + 1, off, cfgb)
            self._write_rest(rows, mt, K, True)
        return mt

    def wurm_multi_rollout(self, foods, heads, bodies, dones, orientations, colours, boost, actions, out_f, out_b, all_done, obs,
                           m, n, N, K, S, T, cfg, seed, call0, off, inj, rinj, stream):
        N, K, S, T, m, n, call0, off = (_int(x) for x in (N, K, S, T, m, n, call0, off))
        self.calls.append('rollout')
        mt = self._rollout(dones, orientations, colours, boost, actions, out_f, out_b, all_done, obs, m, n, N, K, S, T, cfg,
                           call0, off, self.read_m(foods, heads, bodies, N, K, S))
        self.write_m(foods, heads, bodies, N, K, S, mt)
        return 0

    def wurm_multi_rollout_resident(self, foods, heads, bodies, dones, orientations, colours, boost, actions, out_f, out_b,
                                    all_done, obs, m, n, N, K, S, T, cfg, seed, call0, off, resident, valid_addr, lazy, stream):
        N, K, S, T, m, n, call0, off, lazy = (_int(x) for x in (N, K, S, T, m, n, call0, off, lazy))
        valid = ctypes.c_int.from_address(_addr(valid_addr))
        self.calls.append('rollout_resident')
        mir = mem(resident, N, np.uint64)
        if self.rollout_keeps_mirror:
            mt = [int(x) for x in mir] if valid.value else self.read_m(foods, heads, bodies, N, K, S)
            mt = self._rollout(dones, orientations, colours, boost, actions, out_f, out_b, all_done, obs, m, n, N, K, S, T, cfg,
                               call0, off, mt)
            mir[:] = np.array(mt, dtype=np.uint64)
            valid.value = 1
            if not lazy:
                self.write_m(foods, heads, bodies, N, K, S, mt)
        else:   # the library's fallback: a lazy mirror is written out first, the tensors are stepped, the mirror is stale
            if lazy and valid.value:
                self.write_m(foods, heads, bodies, N, K, S, [int(x) for x in mir])
            mt = self._rollout(dones, orientations, colours, boost, actions, out_f, out_b, all_done, obs, m, n, N, K, S, T, cfg,
                               call0, off, self.read_m(foods, heads, bodies, N, K, S))
            self.write_m(foods, heads, bodies, N, K, S, mt)
            valid.value = 0
        return 0


def install(monkeypatch, sim, torch, c_stepper=False, torchinfo=None):
    """points wurm_amd._lib at the simulator (CPU tensors, no device)"""
    monkeypatch.setattr(_lib, 'lib', lambda: sim)
    monkeypatch.setattr(_lib, 'require_device', lambda d: torch.device('cpu'))
    monkeypatch.setattr(_lib, 'stream_ptr', lambda i=None: 0)
    monkeypatch.setattr(_lib, 'call', lambda idx, fn, *a: fn(*a))
    monkeypatch.setattr(_lib, 'accessors', lambda: ((lambda: -1), (lambda i: 0)))
    if isinstance(sim, SimSingle):
        slot = CSlot(sim) if c_stepper else sim.step_slot
        sim._slot_keep = slot
        monkeypatch.setattr(_lib, 'step_slot_fn', lambda name='wurm_single_step_slot': slot)
        monkeypatch.setattr(_lib, 'torch_helpers', lambda: torchinfo)
        import wurm_amd.utils as U

        def consistency_mask(envs):
            e = envs.to(torch.float32).contiguous()
            err = torch.empty(e.shape[0], dtype=torch.int32)
            sim.wurm_single_check(e.data_ptr(), err.data_ptr(), e.shape[0], e.shape[2], 0)
            return err
        monkeypatch.setattr(U, 'consistency_mask', consistency_mask)
    else:
        slot = CMultiSlot(sim) if c_stepper else sim.step_slot
        sim._slot_keep = slot
        monkeypatch.setattr(_lib, 'step_slot_fn', lambda name='wurm_multi_step_slot': slot)
        monkeypatch.setattr(_lib, 'torch_helpers', lambda: torchinfo)
