"""Harness-side shim that makes the *unmodified* reference (/root/reference, pinned to torch==1.1.0 /
Python 3.6) importable under torch 2.10 / Python 3.10 in the build container.

This file is test infrastructure for generating golden fixtures (tests/golden/make_golden.py).  It is
only ever imported in the build container, where /root/reference exists; nothing on the GPU box uses it.
It contains no reference code: it restores the torch-1.1 semantics the reference relies on
(SURVEY.md Appendix B):

  1. collections.Iterable alias                       (reference wurm/utils.py:3)
  2. stub `gym.envs.classic_control.rendering`        (reference wurm/envs/single_snake.py:5)
  3. config.DEFAULT_DEVICE = 'cpu' before wurm import (reference config.py:5)
  4. ~uint8 is a logical not                          (e.g. reference wurm/envs/single_snake.py:246)
  5. uint8 tensors index like masks in __setitem__    (e.g. reference wurm/envs/multi_snake.py:569-572)
  6. int tensor / int is truncating division          (reference wurm/envs/multi_snake.py:276,278)
"""
import collections
import collections.abc
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get('WURM_REFERENCE_ROOT', '/root/reference')

_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    sys.dont_write_bytecode = True

    # 1
    if not hasattr(collections, 'Iterable'):
        collections.Iterable = collections.abc.Iterable

    # 2
    class _Viewer(object):
        isopen = True

        def imshow(self, img):
            pass

    rendering = types.ModuleType('gym.envs.classic_control.rendering')
    rendering.SimpleImageViewer = _Viewer
    classic = types.ModuleType('gym.envs.classic_control')
    classic.rendering = rendering
    envs = types.ModuleType('gym.envs')
    envs.classic_control = classic
    gym = types.ModuleType('gym')
    gym.envs = envs
    sys.modules.setdefault('gym', gym)
    sys.modules.setdefault('gym.envs', envs)
    sys.modules.setdefault('gym.envs.classic_control', classic)
    sys.modules.setdefault('gym.envs.classic_control.rendering', rendering)

    # 3
    # at the END of sys.path: the reference has its own top-level `tests` package, which must not shadow this repo's
    # (spawned gloo workers of tests/test_sharding_gloo.py inherit sys.path and import tests.test_sharding_gloo)
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
    import config
    config.DEFAULT_DEVICE = 'cpu'

    # 4
    _orig_invert = torch.Tensor.__invert__

    def _invert(self):
        if self.dtype == torch.uint8:
            return self == 0
        return _orig_invert(self)

    torch.Tensor.__invert__ = _invert

    # 5
    _orig_setitem = torch.Tensor.__setitem__

    def _fix_index(idx):
        if isinstance(idx, torch.Tensor) and idx.dtype == torch.uint8:
            return idx.bool()
        if isinstance(idx, tuple):
            return tuple(_fix_index(i) for i in idx)
        return idx

    def _setitem(self, idx, value):
        idx = _fix_index(idx)
        if isinstance(value, torch.Tensor) and value.dtype == torch.bool and self.dtype != torch.bool:
            value = value.to(self.dtype)
        return _orig_setitem(self, idx, value)

    torch.Tensor.__setitem__ = _setitem

    # 6
    _orig_truediv = torch.Tensor.__truediv__
    _int_types = (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64)

    def _truediv(self, other):
        if self.dtype in _int_types:
            if isinstance(other, int) or (isinstance(other, torch.Tensor) and other.dtype in _int_types):
                return torch.div(self, other, rounding_mode='trunc')
        return _orig_truediv(self, other)

    torch.Tensor.__truediv__ = _truediv


def import_reference():
    """Returns the reference's (SingleSnake, SimpleGridworld, MultiSnake, wurm.utils) after installing the shim."""
    install()
    from wurm.envs import SingleSnake, SimpleGridworld, MultiSnake
    import wurm.utils as ref_utils
    return SingleSnake, SimpleGridworld, MultiSnake, ref_utils
