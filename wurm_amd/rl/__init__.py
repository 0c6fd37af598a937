"""Learner-side glue that consumes the env's outputs: A2C loss with the return scan on the GPU, and a
preallocated trajectory buffer — same class names as the reference's `wurm.rl`."""
from wurm_amd.rl.a2c import A2C, a2c_returns
from wurm_amd.rl.trajectory_store import TrajectoryStore

__all__ = ['A2C', 'a2c_returns', 'TrajectoryStore']
