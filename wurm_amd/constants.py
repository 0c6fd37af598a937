"""Layout constants of the env state tensors.

The channel indices, the comparison epsilon and the default device keep the names and values callers of the
reference import from its top-level `config` module (reference config.py:5-11), so `from wurm_amd.constants import
BODY_CHANNEL` is the drop-in for `from config import BODY_CHANNEL`.
"""
from enum import IntEnum


class Channel(IntEnum):
    """Channel order of a single-snake state tensor (N, 3, S, S); SimpleGridworld uses the first two."""
    FOOD = 0
    HEAD = 1
    BODY = 2


FOOD_CHANNEL, HEAD_CHANNEL, BODY_CHANNEL = (int(c) for c in Channel)

#: grids hold exact small integers in fp32; the reference compares them against this epsilon
EPS = 1e-6

#: there is no CPU path in this package (see wurm_amd/_lib.py: require_device)
DEFAULT_DEVICE = 'cuda'
