from .a2c import A2C, a2c_returns
from .trajectory_store import TrajectoryStore
