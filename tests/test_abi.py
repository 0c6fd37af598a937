"""CPU checks of the drop-in boundary: libwurm_hip.so loads and exports every entry point include/wurm_hip.h
declares; the host layer maps error codes to the reference's exception types and refuses to run without the
HIP device / library (no silent fallback).  No compute call is made here."""
import ctypes
import os
import re

import pytest

from wurm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'wurm_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(wurm_[a-z_0-9]+)\s*\(', text)))


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'


def test_header_and_loader_agree():
    assert _declared() == sorted(_lib.SYMBOLS)


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f'{name} missing from libwurm_hip.so'
    lib.wurm_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.wurm_version()


def test_obs_elems_queries():
    lib = _lib.lib()
    assert lib.wurm_single_obs_elems(_lib.OBS_PARTIAL, 2, 9) == 75
    assert lib.wurm_single_obs_elems(_lib.OBS_DEFAULT, 0, 36) == 3 * 36 * 36
    assert lib.wurm_single_obs_elems(_lib.OBS_ONE_CHANNEL, 0, 12) == 144
    assert lib.wurm_grid_obs_elems(_lib.OBS_RAW, 0, 9) == 2 * 81
    assert lib.wurm_grid_obs_elems(_lib.OBS_PARTIAL, 2, 9) == 0  # not an observation mode of SimpleGridworld


def test_argument_validation_without_device():
    """Entry points validate before launching: bad arguments return error codes even with no GPU."""
    lib = _lib.lib()
    assert lib.wurm_single_step(None, None, 0, None, None, None, None, None, _lib.OBS_NONE, 0,
                                ctypes.c_int64(4), 9, ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_int64(0),
                                None, None) == _lib.ERR_INVALID_ARG
    assert lib.wurm_single_reset(None, None, None, _lib.OBS_NONE, 0, ctypes.c_int64(0), 8, ctypes.c_uint64(0),
                                 ctypes.c_uint64(0), ctypes.c_int64(0), None, None) == _lib.ERR_UNSUPPORTED
    assert lib.wurm_single_step(None, None, 7, None, None, None, None, None, _lib.OBS_NONE, 0,
                                ctypes.c_int64(0), 9, ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_int64(0),
                                None, None) == _lib.ERR_DTYPE


def test_error_code_mapping():
    _lib.check(0, 'x')
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.ERR_UNSUPPORTED, 'x')
    with pytest.raises(TypeError):
        _lib.check(_lib.ERR_DTYPE, 'x')
    with pytest.raises(RuntimeError):
        _lib.check(_lib.ERR_INVALID_ARG, 'x')
    with pytest.raises(_lib.WurmHipError):
        _lib.check(_lib.ERR_HIP, 'x')


def test_no_cpu_fallback():
    with pytest.raises(_lib.WurmHipError):
        _lib.require_device('cpu')
    import torch
    if not torch.cuda.is_available():
        from wurm_amd.envs import SingleSnake
        with pytest.raises(_lib.WurmHipError):
            SingleSnake(num_envs=2, size=9)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, 'wurm_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src and 'liboracle' not in src, f


def test_obs_mode_parsing():
    assert _lib.parse_obs_mode('partial_3') == (_lib.OBS_PARTIAL, 3)
    assert _lib.parse_obs_mode('default') == (_lib.OBS_DEFAULT, 0)
    with pytest.raises(ValueError):
        _lib.parse_obs_mode('bogus')


def test_resident_mirror_sizes():
    """wurm_single_resident_bytes / wurm_multi_resident_bytes (host arithmetic only): which shapes are offered the mirror
    of the per-call step, from which batch size, and how large it is"""
    l = _lib.lib()
    _lib.set_option('WURM_RESIDENT_MIN_ENVS', None)
    part2, none, default = (_lib.OBS_PARTIAL, 2), (_lib.OBS_NONE, 0), (_lib.OBS_DEFAULT, 0)
    assert l.wurm_single_resident_bytes(65536, 9, *part2) == 65536 * 32          # 9 x 9: 32 bytes per env
    assert l.wurm_single_resident_bytes(4096, 9, *none) == 4096 * 32
    assert l.wurm_single_resident_bytes(4095, 9, *part2) == 0                    # latency-bound anyway
    assert l.wurm_single_resident_bytes(65536, 9, _lib.OBS_PARTIAL, 1) == 65536 * 32   # crops up to 5 x 5,
    assert l.wurm_single_resident_bytes(65536, 9, _lib.OBS_ONE_CHANNEL, 0) == 65536 * 32   # one_channel, default, positions
    assert l.wurm_single_resident_bytes(65536, 9, _lib.OBS_DEFAULT, 0) == 65536 * 32
    assert l.wurm_single_resident_bytes(65536, 9, _lib.OBS_POSITIONS, 0) == 65536 * 32
    assert l.wurm_single_resident_bytes(65536, 9, _lib.OBS_RAW, 0) == 65536 * 32       # (round 5) 'raw': body values from the move queue
    assert l.wurm_single_resident_bytes(65536, 9, _lib.OBS_PARTIAL, 3) == 65536 * 32   # (round 5) 7 x 7 crops through bit planes
    assert l.wurm_single_resident_bytes(65536, 9, _lib.OBS_PARTIAL, 4) == 0            # 9 x 9 crops: the one-env-per-wave kernels
    # (round 6) 10 x 10 / 11 x 11: 48 bytes per env (lane_wide_resident.hpp) for the observations its bit planes serve
    assert l.wurm_single_resident_bytes(65536, 10, *part2) == 65536 * 48 and l.wurm_single_resident_bytes(4096, 11, *default) == 4096 * 48
    assert l.wurm_single_resident_bytes(4095, 10, *part2) == 0
    assert l.wurm_single_resident_bytes(65536, 11, _lib.OBS_RAW, 0) == 0 and l.wurm_single_resident_bytes(65536, 10, _lib.OBS_PARTIAL, 4) == 0
    # 12 x 12 and larger: the 16-bit clock grid (runs of 256 cells) + 48 bytes, every observation mode, from 2^20 cells on
    assert l.wurm_single_resident_bytes(8192, 36, *default) == 8192 * (6 * 512 + 48)
    assert l.wurm_single_resident_bytes(8192, 12, *default) == 8192 * (1 * 512 + 48)
    assert l.wurm_single_resident_bytes(7000, 12, *default) == 0
    assert l.wurm_single_resident_bytes(300, 64, _lib.OBS_PARTIAL, 6) == 300 * (16 * 512 + 48)
    assert l.wurm_single_resident_bytes(1 << 20, 65, *default) == 0
    # MultiSnake: K grids of 16-bit clocks, the food bytes, three ints per snake, each part padded to 16 bytes
    K, S = 4, 25
    per = (2 * K * S * S + 15) // 16 * 16 + (S * S + 15) // 16 * 16 + (12 * K + 15) // 16 * 16
    assert l.wurm_multi_resident_bytes(4096, K, S) == 4096 * per
    assert l.wurm_multi_resident_bytes(100, K, S) == 0
    assert l.wurm_multi_resident_bytes(4096, 65, S) == 0 and l.wurm_multi_resident_bytes(4096, K, 4) == 0
    _lib.set_option('WURM_RESIDENT_MIN_ENVS', 0)                                 # tests / tuning: a number of envs instead
    assert l.wurm_single_resident_bytes(3, 9, *part2) == 96 and l.wurm_single_resident_bytes(3, 12, *default) == 3 * 560
    assert l.wurm_multi_resident_bytes(2, K, S) == 2 * per
    _lib.set_option('WURM_RESIDENT_MIN_ENVS', 1000000000)
    assert l.wurm_single_resident_bytes(65536, 9, *part2) == 0 and l.wurm_multi_resident_bytes(4096, K, S) == 0
    _lib.set_option('WURM_RESIDENT_MIN_ENVS', None)
    # no-op flushes need no device
    assert l.wurm_single_resident_flush(None, None) == _lib.ERR_INVALID_ARG
    c = _lib.SingleCall()
    assert l.wurm_single_resident_flush(ctypes.addressof(c), None) == _lib.OK
    m = _lib.MultiCall()
    assert l.wurm_multi_resident_flush(ctypes.addressof(m), None) == _lib.OK
