// options.hip — the library's knobs (options.hpp): defaults, one read of the environment when the library is loaded,
// wurm_set_option / wurm_get_option afterwards.  Host code only.
#include <atomic>
#include <cerrno>
#include <cstdlib>
#include <cstring>

#include "options.hpp"
#include "../../include/wurm_hip.h"

namespace wurm {

constexpr Options DEFAULTS = {1ll << 20, 12288, 6144, -1, -1, -1, 12, 0, 2048, 0, -1, 0, 0, 1, -1, -1};
// PROCESS-WIDE: every env object and every thread of the process sees the same knobs (documented in include/wurm_hip.h);
// the launch counter is a diagnostic that concurrent launchers may bump at the same time, hence atomic.
Options opt = DEFAULTS;
std::atomic<long long> launch_count{0};

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
    {"WURM_MULTI_SHAPE_KERNELS", &Options::multi_shape_kernels},
    {"WURM_GRIDWORLD_LANE_EPW", &Options::gridworld_lane_epw},
    {"WURM_GRID_ROLLOUT_MIN_SIZE", &Options::grid_rollout_min_size},
};

const Entry *find(const char *name)
{
    if (!name) return nullptr;
    for (const Entry &e : table)
        if (strcmp(e.name, name) == 0) return &e;
    return nullptr;
}

// an environment variable that is not a whole decimal number (empty, garbage, out of range) leaves the default in place —
// atoll() would have turned it into 0, which for the *_MIN_ENVS thresholds means "always"
bool parse_whole(const char *v, long long &out)
{
    if (!v || !*v) return false;
    errno = 0;
    char *end = nullptr;
    const long long x = strtoll(v, &end, 10);
    if (errno != 0 || end == v) return false;
    while (*end == ' ' || *end == '\t' || *end == '\n') ++end;
    if (*end != '\0') return false;
    out = x;
    return true;
}

__attribute__((constructor)) void read_environment()
{
    for (const Entry &e : table) {
        long long x;
        if (parse_whole(getenv(e.name), x)) opt.*(e.field) = x;
    }
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

int64_t wurm_launch_count(void) { return (int64_t)wurm::launch_count.load(std::memory_order_relaxed); }

int wurm_reset_option(const char *name)
{
    const wurm::Entry *e = wurm::find(name);
    if (!e) return WURM_ERR_INVALID_ARG;
    wurm::opt.*(e->field) = wurm::DEFAULTS.*(e->field);
    return WURM_OK;
}

} // extern "C"
