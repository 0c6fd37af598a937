"""one launch shape of the SimpleGridworld lane rollout under rocprofv3 --kernel-trace --stats (tools/gridworld_probe.py times it with events)"""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from wurm_amd.envs import SimpleGridworld
from wurm_amd import _lib
mode, T, epw, var = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device('cuda:0')
with _lib.knobs(WURM_GRIDWORLD_LANE_EPW=epw, WURM_GRID_ROTATE=var):
    env = SimpleGridworld(65536, 9, start_location=(4, 4), observation_mode=mode, device=dev, seed=0)
    acts = torch.randint(4, (9, T, 65536), device=dev)
    for i in range(9):
        env.rollout(acts[i])
    torch.cuda.synchronize()
