// options.hip — the library's knobs (options.hpp): defaults, one read of the environment when the library is loaded,
// wurm_set_option / wurm_get_option afterwards.  Host code only.
#include <cstdlib>
#include <cstring>

#include "options.hpp"
#include "../../include/wurm_hip.h"

namespace wurm {

constexpr Options DEFAULTS = {1ll << 20, 12288, 6144, -1, -1, -1, 12, 0, 2048, 0, -1, 0, 0};
Options opt = DEFAULTS;
long long launch_count = 0;

namespace {

struct Entry {
    const char *name;
    long long Options::*field;
};

const Entry table[] = {
    {"WURM_GRID_STEP_MIN_CELLS", &Options::grid_step_min_cells},
    {"WURM_LANE_STEP_MIN_ENVS", &Options::lane_step_min_envs},
    {"WURM_LANE_ROLLOUT_MIN_ENVS", &Options::lane_rollout_min_envs},
    {"WURM_LANE_ROLLOUT_EPW", &Options::lane_rollout_epw},
    {"WURM_RESIDENT_MIN_ENVS", &Options::resident_min_envs},
    {"WURM_RESIDENT_EPW", &Options::resident_epw},
    {"WURM_GRID_WAVES_PER_CU", &Options::grid_waves_per_cu},
    {"WURM_POLICY_GENERIC", &Options::policy_generic},
    {"WURM_MULTI_GROUP_MIN_ENVS", &Options::multi_group_min_envs},
    {"WURM_MULTI_GROUP_VARIANT", &Options::multi_group_variant},
    {"WURM_MULTI_GROUP_STEP_WPB", &Options::multi_group_step_wpb},
    {"WURM_MULTI_GROUP_SHAPE", &Options::multi_group_shape},
    {"WURM_GRID_ROTATE", &Options::grid_rotate},
};

const Entry *find(const char *name)
{
    if (!name) return nullptr;
    for (const Entry &e : table)
        if (strcmp(e.name, name) == 0) return &e;
    return nullptr;
}

__attribute__((constructor)) void read_environment()
{
    for (const Entry &e : table)
        if (const char *v = getenv(e.name)) opt.*(e.field) = atoll(v);
}

} // namespace
} // namespace wurm

extern "C" {

int wurm_set_option(const char *name, int64_t value)
{
    const wurm::Entry *e = wurm::find(name);
    if (!e) return WURM_ERR_INVALID_ARG;
    wurm::opt.*(e->field) = (long long)value;
    return WURM_OK;
}

int64_t wurm_get_option(const char *name)
{
    const wurm::Entry *e = wurm::find(name);
    return e ? (int64_t)(wurm::opt.*(e->field)) : INT64_MIN;
}

int64_t wurm_launch_count(void) { return (int64_t)wurm::launch_count; }

int wurm_reset_option(const char *name)
{
    const wurm::Entry *e = wurm::find(name);
    if (!e) return WURM_ERR_INVALID_ARG;
    wurm::opt.*(e->field) = wurm::DEFAULTS.*(e->field);
    return WURM_OK;
}

} // extern "C"
