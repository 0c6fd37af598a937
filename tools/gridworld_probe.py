import torch, time, sys
sys.path.insert(0, '/root/repo')
from wurm_amd.envs import SimpleGridworld
from wurm_amd import _lib
dev = torch.device('cuda:0')
for mode, T in (('default', 16), ('raw', 16), ('positions', 64)):
    for knob, epw, var in ((0, -1, 0), (0, -1, 1), (0, 32, 0), (0, 32, 1), (0, -1, 0), (0, -1, 1), (1 << 40, -1, 0)):
        if mode == 'positions' and epw not in (64, -1):
            continue
        with _lib.knobs(WURM_LANE_ROLLOUT_MIN_ENVS=knob, WURM_GRIDWORLD_LANE_EPW=epw, WURM_GRID_ROTATE=var):
            env = SimpleGridworld(65536, 9, start_location=(4, 4), observation_mode=mode, device=dev, seed=0)
            acts = torch.randint(4, (9, T, 65536), device=dev)
            env.rollout(acts[0]); torch.cuda.synchronize()
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(1, 9): env.rollout(acts[i])
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 8)
            ts.sort()
            print(mode, T, 'lane epw %d var %d' % (epw, var) if knob == 0 else 'generic', _lib.lib().wurm_single_last_route().decode(),
                  'ms %.4f' % ts[2], 'eps %.3e' % (65536 * T / ts[2] * 1e3), flush=True)
