// step_args.hpp — the kernel argument block and action accessors shared by the SingleSnake / SimpleGridworld
// translation units (single_snake.hip, grid_rollout.hip).
#pragma once

#include "wurm_device.hpp"
#include "../../include/wurm_hip.h"

namespace wurm {

struct StepArgs {
    float *envs;
    void *actions;
    int act_dtype;
    float *reward;
    uint8_t *done, *selfc, *edgec;
    float *obs;
    int obs_mode, obs_n;
    long long obs_elems;
    long long N;
    int S;
    long long T;
    int start_y, start_x;
    u64 seed, call;
    long long env_offset;
    const int *inject_food;
    const int *inject_reset;
    const uint8_t *done_in;
    int lds_per_wave;
    // fused step (+ deferred / immediate reset): wurm_single_step_reset
    float *obs_after;
    uint8_t *done_copy;
    const int *inject_pre_reset;
    u64 pre_call;
    int post_reset;
    int only_flagged; // rollout_kernel: process only the envs the grid kernel marked GRID_SKIPPED in done[0][env]
    int grid_rotate;  // grid_rollout.hip: every env's observation rows start at an env-dependent row (make_grid)
    // grid_step_kernel: the caller's compact mirror of the state (wurm_single_call.resident, S >= 12: the clock grids
    // and the per-env scalars as the kernel holds them), nullable; valid: it describes envs; lazy: envs are not written
    void *resident;
    int resident_valid, resident_lazy;
};

__device__ __forceinline__ long long load_action(const void *actions, int dtype, long long i)
{
    return dtype == WURM_ACT_I64 ? ((const long long *)actions)[i] : (long long)((const int *)actions)[i];
}

__device__ __forceinline__ void store_action(void *actions, int dtype, long long i, long long v)
{
    if (dtype == WURM_ACT_I64) ((long long *)actions)[i] = v;
    else ((int *)actions)[i] = (int)v;
}

// LDS-resident clock-grid rollout (grid_rollout.hip): true if it took the launch (SingleSnake, S >= 12).  Envs whose
// state is outside its domain are left untouched and marked with done[0][env] = GRID_SKIPPED for the generic kernel.
constexpr uint8_t GRID_SKIPPED = 0xFF;
bool grid_rollout_eligible(const StepArgs &p);
hipError_t launch_grid_rollout(const StepArgs &p, hipStream_t stream);
// the same for one per-call iteration (fused_step_kernel's contract)
bool grid_step_eligible(const StepArgs &p);
// gridworld_lane.hip — SimpleGridworld rollouts of large batches, one env per lane; envs outside its domain are marked
// GRID_SKIPPED in the same way
bool gridworld_lane_eligible(const StepArgs &p);
hipError_t launch_gridworld_lane_rollout(const StepArgs &p, hipStream_t stream);
// ... and the per-call step of large batches under fused_step_kernel's contract (K_STEP / K_FUSED)
bool gridworld_lane_step_eligible(const StepArgs &p);
hipError_t launch_gridworld_lane_step(const StepArgs &p, hipStream_t stream);
// ... on the caller's mirror of the state (p.resident: a header + one 32-bit record per env, gridworld_lane.hip)
long long gridworld_resident_bytes(long long N, int S, int obs_mode, long long obs_elems);
hipError_t launch_gridworld_lane_flush(const StepArgs &p, hipStream_t stream);
hipError_t launch_grid_step(const StepArgs &p, hipStream_t stream);

// one-env-per-LANE rollout for large batches of 9 x 9 SingleSnake (lane_rollout.hip / lane_rollout.hpp); envs outside its
// domain are rolled out by the one-env-per-wave code inside the same launch
bool lane_rollout_eligible(const StepArgs &p);
hipError_t launch_lane_rollout(const StepArgs &p, hipStream_t stream);

// ... and for 10 x 10 / 11 x 11 (lane_wide.hip / lane_wide.hpp: 128-bit occupancy masks)
bool lane_wide_eligible(const StepArgs &p);
hipError_t launch_lane_wide(const StepArgs &p, hipStream_t stream);

// ... and their per-call step on a caller-owned mirror (lane_wide_resident.hpp; lazy form only)
bool lane_wide_resident_shape(int S, int obs_mode, int obs_n);
bool lane_wide_resident_eligible(const StepArgs &p);
hipError_t launch_lane_wide_resident(const StepArgs &p, void *resident, bool valid, uint32_t *check_mask, hipStream_t stream);
hipError_t launch_lane_wide_resident_flush(const StepArgs &p, void *resident, hipStream_t stream);

// per-call step of large 9 x 9 batches on a caller-owned compact mirror of the state (lane_resident.hpp, in lane_rollout.hip)
bool lane_resident_shape(int S, int obs_mode, int obs_n);
bool lane_resident_eligible(const StepArgs &p);
hipError_t launch_lane_resident(const StepArgs &p, void *resident, bool valid, bool lazy, uint32_t *check_mask, hipStream_t stream);
hipError_t launch_lane_resident_flush(const StepArgs &p, void *resident, hipStream_t stream);
// the same for grids of 12 x 12 and larger (grid_rollout.hip: grid_step_kernel reads / maintains p.resident)
bool grid_resident_eligible(const StepArgs &p);
long long grid_resident_bytes(long long N, int S);
hipError_t launch_grid_resident_flush(const StepArgs &p, hipStream_t stream);

} // namespace wurm
