"""Batched environments with the class API of the reference's `wurm.envs`, running on the gfx950 kernels behind
include/wurm_hip.h."""
from wurm_amd.envs.multi_snake import MultiSnake
from wurm_amd.envs.simple_gridworld import SimpleGridworld
from wurm_amd.envs.single_snake import SingleSnake

__all__ = ['SingleSnake', 'SimpleGridworld', 'MultiSnake']
