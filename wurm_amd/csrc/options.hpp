// options.hpp — tuning / test knobs of the library, read ONCE (at library load, from the environment) and changed
// afterwards only through wurm_set_option (include/wurm_hip.h).  Launch paths read plain fields of `wurm::opt`: no
// getenv on any path whose whole budget is a few microseconds.
#pragma once
#include <atomic>

namespace wurm {

struct Options {
    long long grid_step_min_cells;   // WURM_GRID_STEP_MIN_CELLS   per-call LDS clock-grid step from this many cells (2^20)
    long long lane_step_min_envs;    // WURM_LANE_STEP_MIN_ENVS    per-call one-env-per-lane step from this many envs (12288)
    long long lane_rollout_min_envs; // WURM_LANE_ROLLOUT_MIN_ENVS one-env-per-lane rollout from this many envs (6144)
    long long lane_rollout_epw;      // WURM_LANE_ROLLOUT_EPW      envs per wave of that rollout (-1 = by batch size)
    long long resident_min_envs;     // WURM_RESIDENT_MIN_ENVS     resident mirror from this many envs (-1 = by shape)
    long long resident_epw;          // WURM_RESIDENT_EPW          envs per wave of the resident step (-1 = by batch size)
    long long grid_waves_per_cu;     // WURM_GRID_WAVES_PER_CU     residency of the clock-grid rollout (12)
    long long policy_generic;        // WURM_POLICY_GENERIC        1 = fused actor on the generic loop even on 9 x 9
    long long multi_group_min_envs;  // WURM_MULTI_GROUP_MIN_ENVS  MultiSnake 'full' rollout: grouped writer from this many envs
    long long multi_group_variant;   // WURM_MULTI_GROUP_VARIANT   tuning / probe bits (multi_snake.hip: grp_variant; none changes a result in the shipped build)
    long long multi_group_step_wpb;  // WURM_MULTI_GROUP_STEP_WPB  per-call step: envs per workgroup of the grouped writer (-1 auto, 0 off)
    long long multi_group_shape;     // WURM_MULTI_GROUP_SHAPE     100 G + 10 W + waves per SIMD of that kernel (0 = automatic)
    long long grid_rotate;           // WURM_GRID_ROTATE           clock-grid kernels: observation rows start at an env-dependent row (0 = off; 1: env % rows, k >= 2: (env % k) * rows / k; measured no better)
    long long multi_shape_kernels;   // WURM_MULTI_SHAPE_KERNELS   MultiSnake: kernels with K, S and the crop radius compiled in for the reference's experiment shapes (1; 0 = the generic kernels)
    long long gridworld_lane_epw;    // WURM_GRIDWORLD_LANE_EPW    SimpleGridworld lane rollout, image modes: envs per wave (4..64; -1 = by batch size)
    long long grid_rollout_min_size; // WURM_GRID_ROLLOUT_MIN_SIZE SingleSnake rollouts on LDS clock grids from this grid size on (-1 = by observation mode: 14 / 18 / 26; 12 = every size they serve)
};

extern Options opt;

// number of kernels this library has launched in this process (wurm_launch_count, include/wurm_hip.h): bench.py reports
// launches per loop iteration from it.  Atomic: concurrent launchers may bump it at the same time.
extern std::atomic<long long> launch_count;
#define WURM_LAUNCH(...) do { ::wurm::launch_count.fetch_add(1, std::memory_order_relaxed); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

} // namespace wurm
