// lane_rollout.hip — translation unit of the one-env-per-LANE rollout (lane_rollout.hpp) and of the per-call step on a
// resident compact state built from the same pieces (lane_resident.hpp).  It needs the device code of
// single_snake.hip (state load / store, rollout_generic for envs outside its domain, the 9 x 9 reset draw) and none of its
// kernels or entry points, so it includes that file with the host side switched off; compiled on its own it builds in a
// fraction of the time of single_snake.hip.
#define WURM_SINGLE_SNAKE_DEVICE_CODE_ONLY
#include "single_snake.hip"
#include "lane_rollout.hpp"
#include "lane_resident.hpp"
