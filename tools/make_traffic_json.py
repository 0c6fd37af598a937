#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a parse_pmc.py summary: per-launch HBM bytes of the env kernels, corrected with
the calibration launches of tools/traffic_workload.py (MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are KiB;
gfx950 FETCH_SIZE under-reports streaming reads — the factor is measured on a known byte count in this access
pattern rather than assumed)."""
import json
import sys

s = json.load(open(sys.argv[1]))


def find(prefix, grid):
    for k, v in s.items():
        if k.startswith(prefix) and k.endswith(f'grid={grid}'):
            return v
    raise KeyError((prefix, grid))


known_read = 65536 * 3 * 36 * 36 * 4  # bytes read by check_kernel<24> over the 65536 x 36 x 36 state
cal = find('void wurm::check_kernel<24>', 4194304)
read_factor = known_read / (cal['FETCH_SIZE']['mean'] * 1024)
known_write = 65536 * 3 * 36 * 36 * 4  # bytes written by reset_kernel<24> rebuilding every env
calw = find('void wurm::reset_kernel<24, true>', 4194304)
write_factor = known_write / (calw['WRITE_SIZE']['mean'] * 1024)
copy = find('__amd_rocclr_copyBuffer', 131072)


def traffic(prefix, grid):
    try:
        v = find(prefix, grid)
    except KeyError:
        return None
    r = v['FETCH_SIZE']['mean'] * 1024 * read_factor
    w = v['WRITE_SIZE']['mean'] * 1024 * write_factor
    return {'read_bytes': r, 'write_bytes': w, 'total_bytes': r + w, 'launches': v['FETCH_SIZE']['launches']}


def total(t):
    return t['total_bytes'] if t else None


detail = {
    'rollout_512x9_chunk1024': traffic('void wurm::rollout_s9_kernel<4', 32768),
    'rollout_8192x9_chunk128': traffic('void wurm::rollout_s9_kernel<4', 524288),
    'rollout_16384x9_chunk128': traffic('void wurm::rollout_s9_kernel<4', 1048576),
    'rollout_32768x9_chunk64': traffic('void wurm::rollout_s9_kernel<4', 2097152),
    'rollout_65536x9_chunk64': traffic('void wurm::rollout_s9_kernel<4', 4194304),
    'rollout_cfg5_8192x36_default_chunk16': traffic('void wurm::(anonymous namespace)::grid_rollout_kernel<true>', 524288),
    'multi_rollout_cfg4_4096x25_k4_full_chunk16': traffic('void wurm::multi_rollout_kernel<true>', 524288),
    'fused_step_512x9_partial2': traffic('void wurm::fused_step_kernel<2, true>', 32768),
    'fused_step_8192x9_partial2': traffic('void wurm::fused_step_kernel<2, true>', 524288),
    'fused_step_65536x9_partial2': traffic('void wurm::fused_step_kernel<2, true>', 4194304),
    'lane_step_65536x9_partial2': traffic('void wurm::lane_step_kernel<16, 9>', 262144),
    'multi_step_cfg4_4096x25_k4_full': traffic('wurm::multi_step_kernel', 262144),
    'grid_step_8192x36_default': traffic('void wurm::(anonymous namespace)::grid_step_kernel<true>', 524288),
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
