"""wurm_amd — MI355X-native (gfx950) implementation of the batched environment step of oscarknagg/wurm.

`wurm_amd.envs` mirrors `wurm.envs` (SingleSnake, SimpleGridworld, MultiSnake); the step / reset / observation
functions run as hand-written HIP kernels behind the C ABI declared in include/wurm_hip.h.
"""
__version__ = '0.1'
