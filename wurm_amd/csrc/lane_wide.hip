// lane_wide.hip — translation unit of the one-env-per-LANE rollout of 10 x 10 and 11 x 11 SingleSnake grids (lane_wide.hpp) and of
// the per-call step of those sizes on a resident compact state (lane_wide_resident.hpp).
// Like lane_rollout.hip it needs the device code of single_snake.hip (state load / store, rollout_generic for envs outside
// its domain, the reset draw) and none of its kernels or entry points.
#define WURM_SINGLE_SNAKE_DEVICE_CODE_ONLY
#include "single_snake.hip"
#include "lane_wide.hpp"
#include "lane_wide_resident.hpp"
