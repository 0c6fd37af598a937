#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a parse_pmc.py summary: per-launch HBM bytes of the env kernels, corrected with
the calibration launches of tools/traffic_workload.py (MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are KiB;
gfx950 FETCH_SIZE under-reports streaming reads — the factor is measured on a known byte count in this access
pattern rather than assumed)."""
import json
import sys

s = json.load(open(sys.argv[1]))


def find(prefix, grid, part=None):
    """the summary entry of the kernel whose name starts with `prefix` at that grid size (threads); part = (i, n): the i-th
    of n equal runs of launches in dispatch order — tools/traffic_workload.py launches one kernel at one grid size for
    more than one configuration (MultiSnake 4 x 25 x 25 and 10 x 36 x 36, both 4096 envs)"""
    hits = [(k, v) for k, v in s.items() if k.startswith(prefix) and k.endswith(f'|grid={grid}')]
    if len(hits) != 1:
        raise KeyError((prefix, grid, [k for k, _ in hits]))
    v = hits[0][1]
    if part is None:
        return v
    i, n = part
    out = {}
    for c, x in v.items():
        vals = x['values']
        m = len(vals) // n
        sub = vals[i * m:(i + 1) * m]
        out[c] = {'launches': len(sub), 'mean': sum(sub) / len(sub), 'min': min(sub), 'max': max(sub)}
    return out


known_read = 65536 * 3 * 36 * 36 * 4  # bytes read by check_kernel<24> over the 65536 x 36 x 36 state
cal = find('void wurm::check_kernel<24>', 4194304)
read_factor = known_read / (cal['FETCH_SIZE']['mean'] * 1024)
known_write = 65536 * 3 * 36 * 36 * 4  # bytes written by reset_kernel<24> rebuilding every env
calw = find('void wurm::reset_kernel<24, true>', 4194304)
write_factor = known_write / (calw['WRITE_SIZE']['mean'] * 1024)
copy = find('__amd_rocclr_copyBuffer', 131072)


def traffic(prefix, grid, part=None):
    try:
        v = find(prefix, grid, part)
    except KeyError:
        return None
    r = v['FETCH_SIZE']['mean'] * 1024 * read_factor
    w = v['WRITE_SIZE']['mean'] * 1024 * write_factor
    return {'read_bytes': r, 'write_bytes': w, 'total_bytes': r + w, 'launches': v['FETCH_SIZE']['launches']}


def total(t):
    return t['total_bytes'] if t else None


detail = {
    'rollout_512x9_chunk1024': traffic('void wurm::rollout_s9_kernel<4', 32768),
    # one env per LANE from 6 144 envs on (lane_rollout.hpp): 8 / 16 / 32 / 64 envs per wave at these batch sizes (round 6's sweep)
    'rollout_8192x9_chunk128': traffic('void wurm::lane_rollout_kernel<8, 4, false>', 65536),
    'rollout_16384x9_chunk128': traffic('void wurm::lane_rollout_kernel<16, 4, false>', 65536),
    'rollout_32768x9_chunk64': traffic('void wurm::lane_rollout_kernel<32, 4, false>', 65536),
    'rollout_65536x9_chunk64': traffic('void wurm::lane_rollout_kernel<64, 4, false>', 65536),
    'rollout_cfg5_8192x36_default_chunk16': traffic('void wurm::(anonymous namespace)::grid_rollout_kernel<true>', 524288),
    # round 4: G envs per workgroup (multi_rollout_group_kernel): cfg4 512 workgroups of 8 steppers + 4 writers, the speeds.py
    # shape 1024 workgroups of 4 + 10
    # (round 6: 4 envs + 4 writers per workgroup with K and S compiled in: 1024 workgroups of 512 threads)
    'multi_rollout_cfg4_4096x25_k4_full_chunk16': traffic('void wurm::multi_rollout_group_kernel<4, 4, 1, 4, false, false, 4, 25>', 524288) or
                                                  traffic('void wurm::multi_rollout_group_kernel<8, 2, 1, 5, false, false', 327680) or
                                                  traffic('void wurm::multi_rollout_group_kernel<8, 4, 1, 6, false, false', 393216),
    'multi_rollout_speeds_4096x36_k10_chunk4': traffic('void wurm::multi_rollout_group_kernel<4, 10, 1, 4, true, false', 917504),
    # round 4: one_channel / default of 65 536 x 9 x 9 through the lane kernels (bit planes): rollout, and per call (reference form)
    'rollout_65536x9_one_channel_chunk32': traffic('void wurm::lane_rollout_kernel<64, -2, false>', 65536),
    'rollout_65536x9_default_chunk32': traffic('void wurm::lane_rollout_kernel<64, -3, false>', 65536),
    'resident_step_65536x9_one_channel_reset_obs': traffic('void wurm::lane_resident_step_kernel<32, 2, -2, true>', 131072),
    'resident_step_65536x9_default_reset_obs': traffic('void wurm::lane_resident_step_kernel<32, 2, -3, true>', 131072),
    'rollout_65536x9_raw_chunk32': traffic('void wurm::lane_rollout_kernel<64, -5, false>', 65536),
    'rollout_65536x9_partial_3_chunk32': traffic('void wurm::lane_rollout_kernel<64, -4, false>', 65536),
    # round 6: 10 x 10 / 11 x 11 one env per lane (lane_wide.hpp), 32 envs per wave at this batch size
    'rollout_65536x10_partial_2_chunk32': traffic('void wurm::lane_wide_rollout_kernel<32, 10, 4, 5, false>', 131072),
    'rollout_65536x11_default_chunk32': traffic('void wurm::lane_wide_rollout_kernel<32, 11, -3, 0, false>', 131072),
    # round 5: SimpleGridworld one env per lane (32 envs per wave at this batch size): zero fill + two floats per env
    # (round 6: 64 envs per wave at this batch size, the run composed as a bit string)
    'rollout_65536x9_gridworld_default_chunk16': traffic('void wurm::(anonymous namespace)::gridworld_lane_rollout_kernel<0, 64>', 65536),
    'rollout_65536x9_gridworld_raw_chunk16': traffic('void wurm::(anonymous namespace)::gridworld_lane_rollout_kernel<1, 64>', 65536),
    # round 6: the per-call step of SimpleGridworld on its mirror (one record per env), reference form (two observations per call)
    'gridworld_step_65536x9_default_reset_obs': traffic('void wurm::(anonymous namespace)::gridworld_lane_step_kernel<0, 32>', 131072),
    # (round 6: K, S and the crop radius compiled in — multi_rollout_kernel<false, false, 4, 4, 25, 5>)
    'multi_rollout_cfg4prime_4096x25_k4_partial5_chunk16': traffic('void wurm::multi_rollout_kernel<false, false, 4', 262144),
    'fused_step_512x9_partial2': traffic('void wurm::fused_step_kernel<2, true>', 32768),
    'fused_step_8192x9_partial2': traffic('void wurm::fused_step_kernel<2, true>', 524288),
    'fused_step_65536x9_partial2': traffic('void wurm::fused_step_kernel<2, true>', 4194304),
    'lane_step_65536x9_partial2': traffic('void wurm::lane_step_kernel<16, 9>', 262144),
    # the per-call step on the resident mirror (lane_resident.hpp, lazy form): without / with the reset observation
    'resident_step_8192x9_partial2': traffic('void wurm::lane_resident_step_kernel<16, 1, 4, true>', 32768),
    'resident_step_65536x9_partial2': traffic('void wurm::lane_resident_step_kernel<32, 1, 4, true>', 131072),
    'resident_step_65536x9_partial2_reset_obs': traffic('void wurm::lane_resident_step_kernel<32, 2, 4, true>', 131072),
    # multi_step_kernel<INJ, OBS> (round 5: one instantiation per observation family).  Dispatch order of <false, 0> ('full')
    # at this grid: 20 launches of cfg4 on the resident mirror (lazy), 20 with the mirror switched off; the second ten of
    # each.  <false, 4> (partial_n): the 20 launches of cfg4', the second ten
    'multi_step_cfg4_4096x25_k4_full': traffic('void wurm::multi_step_kernel<false, 0', 262144, (1, 4)),
    'multi_step_cfg4_4096x25_k4_full_no_mirror': traffic('void wurm::multi_step_kernel<false, 0', 262144, (3, 4)),
    'per_call_api_cfg4prime_4096x25_k4_partial5': traffic('void wurm::multi_step_kernel<false, 4', 262144, (1, 2)),
    'per_call_api_speeds_4096x36_k10': traffic('void wurm::multi_step_wg_kernel<false', 1048576) or traffic('wurm::multi_step_wg_kernel', 1048576),
    # dispatch order: 30 launches on the resident mirror (lazy: the fp32 state is neither read nor written), then 30 without
    'grid_step_8192x36_default': traffic('void wurm::(anonymous namespace)::grid_step_kernel<true>', 524288, (0, 2)),
    'grid_step_8192x36_default_no_mirror': traffic('void wurm::(anonymous namespace)::grid_step_kernel<true>', 524288, (1, 2)),
}
out = {
    '_calibration': {
        'read_factor_dword_per_lane': read_factor, 'write_factor_dword_per_lane': write_factor,
        'copy_1GiB_FETCH_SIZE_KiB': copy['FETCH_SIZE']['mean'], 'copy_1GiB_WRITE_SIZE_KiB': copy['WRITE_SIZE']['mean'],
        'note': 'FETCH_SIZE reads 1/2 of the true bytes for 16 B/lane copies AND for the kernels\' dword-per-lane '
                'coalesced reads; WRITE_SIZE is exact for both.'},
    # the keys bench.py looks up: rollout_<N>x<S>_chunk<steps per launch>
    'rollout_512x9_chunk1024': total(detail['rollout_512x9_chunk1024']),
    'rollout_8192x9_chunk128': total(detail['rollout_8192x9_chunk128']),
    'rollout_16384x9_chunk128': total(detail['rollout_16384x9_chunk128']),
    'rollout_32768x9_chunk64': total(detail['rollout_32768x9_chunk64']),
    'rollout_65536x9_chunk64': total(detail['rollout_65536x9_chunk64']),
    'detail': detail,
}
print(json.dumps(out, indent=1))
