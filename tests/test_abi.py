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
