/*
 * wurm_hip.h — C ABI of libwurm_hip.so, the MI355X (gfx950) implementation of the batched environment step of
 * oscarknagg/wurm: wurm.envs.SingleSnake / SimpleGridworld / MultiSnake .step() / .reset() / ._observe().
 *
 * The reference exposes no FFI for this path — its boundary is the Python class API of wurm/envs
 * (wurm/envs/__init__.py:1-3).  Each entry point below names the reference method it replaces; the Python
 * classes in wurm_amd/envs/ (same constructor signatures, attributes and return conventions as the
 * reference's) are thin callers of this ABI, and INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add to call it from the wurm/envs modules directly.
 *
 * Conventions
 *  - All pointers are DEVICE pointers (HBM), caller-allocated and caller-owned.  Nothing is allocated, freed
 *    or synchronised here: each call enqueues kernels on `stream` (a hipStream_t passed as void*, NULL = the
 *    default stream) and returns.
 *  - State layout is the reference's: fp32 NCHW holding exact small integers.
 *      SingleSnake     envs (N,3,S,S) = [food, head, body]         (single_snake.py:87, config.py:7-9)
 *      SimpleGridworld envs (N,2,S,S) = [food, agent]              (simple_gridworld.py:74)
 *      MultiSnake      foods (N,1,S,S), heads/bodies (N*K,1,S,S), agent = env*K + i   (multi_snake.py:100-108)
 *  - done / info outputs are one byte per env holding 0 or 1 (torch.bool / torch.uint8 storage).
 *  - Return value: WURM_OK or a negative WURM_ERR_* code; the Python layer maps codes onto the reference's
 *    exception types (TypeError / RuntimeError / NotImplementedError / ValueError).
 *  - Randomness: the reference draws from torch's global RNG in a way no other implementation can restate
 *    (SURVEY.md §0 fact 6).  This ABI uses a counter-based generator, Philox4x32-10 keyed by `seed`, with
 *    counter (global env id = env_offset + i, `call`, purpose): results do not depend on how envs are
 *    sharded over GPUs.  `call` must be different for every step()/reset() call of one env object (the
 *    Python classes count calls).  Every `inject_*` argument is nullable; when non-NULL it supplies the
 *    random outcomes (used to replay outcomes recorded from the reference in parity tests).
 *  - Supported grid sizes: 3 <= S <= 64.  Env ids must be < 2^32.
 */
#ifndef WURM_HIP_H
#define WURM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WURM_OK 0
#define WURM_ERR_INVALID_ARG (-1) /* -> RuntimeError / ValueError   */
#define WURM_ERR_UNSUPPORTED (-2) /* -> NotImplementedError         */
#define WURM_ERR_HIP (-3)         /* a HIP launch failed            */
#define WURM_ERR_DTYPE (-4)       /* -> TypeError                   */
#define WURM_MIRROR_REFUSED 1     /* wurm_grid_step_reset only: the call ran; the mirror it was asked to build was refused */

/* observation modes: single_snake.py:130-195, simple_gridworld.py:111-133, multi_snake.py:283-334 */
#define WURM_OBS_DEFAULT 0     /* RGB/255, (N,3,S,S)  (MultiSnake: 'full', per agent)         */
#define WURM_OBS_RAW 1         /* clone of the state                                          */
#define WURM_OBS_ONE_CHANNEL 2 /* (N,1,S,S)                                                   */
#define WURM_OBS_POSITIONS 3   /* (N,4) head y,x food y,x                                     */
#define WURM_OBS_PARTIAL 4     /* (N,3*(2n+1)^2) crop around the head                         */
#define WURM_OBS_NONE 5        /* no observation written                                      */

/* element type of the `actions` tensor (single_snake.py:198-200 accepts short/int/long) */
#define WURM_ACT_I64 0
#define WURM_ACT_I32 1

/* Library / build identification: "wurm_hip <version> gfx950". */
const char *wurm_version(void);

/* Tuning / test knobs (wurm_amd/csrc/options.hpp lists them: WURM_LANE_ROLLOUT_MIN_ENVS, WURM_RESIDENT_MIN_ENVS, ...).
 * Each starts at its default, is overridden ONCE by the environment variable of the same name when the library is
 * loaded, and changes afterwards only through these calls — no launch path reads the environment.  None of them
 * changes results, only which kernel serves a call.  set / reset return WURM_ERR_INVALID_ARG for an unknown name,
 * get returns INT64_MIN.  The knobs are PROCESS-WIDE (every env object and thread sees them); an environment variable that
 * is not a whole decimal number leaves the default in place.  wurm_launch_count is an atomic diagnostic counter,
 * wurm_single_last_route names the calling thread's last launch; neither is state a later call depends on. */
int wurm_set_option(const char *name, int64_t value);
int64_t wurm_get_option(const char *name);
int wurm_reset_option(const char *name);

/* Number of kernels the library has launched in this process so far (a plain counter: bench.py divides its growth by
 * the loop iterations to report launches per `step(a); reset(done)` iteration). */
int64_t wurm_launch_count(void);

/* Name of the row of the dispatch table (wurm_amd/csrc/single_snake.hip: route_of) that served the last SingleSnake /
 * SimpleGridworld launch of this process: "generic", "grid_step", "lane_step", "lane_resident", "lane_wide_resident", "lane_wide", "grid_rollout",
 * "lane_rollout", "rollout_s9", "rollout_s9_injected", "rollout_lean", "rollout_generic_partial", "rollout_generic_none".
 * Diagnostic only (bench.py and tests/test_dispatch_table.py name a launch by it); a static string. */
const char *wurm_single_last_route(void);

/* Number of fp32 elements one env's observation occupies (0 = invalid mode for that env family). */
int64_t wurm_single_obs_elems(int obs_mode, int obs_n, int size);
int64_t wurm_grid_obs_elems(int obs_mode, int obs_n, int size);

/* ------------------------------------------------------------------------------------------- SingleSnake */

/* SingleSnake.step (single_snake.py:197-304) + its _observe (:130-195) in one launch.
 *   envs            in/out (N,3,S,S) fp32
 *   actions         in/out (N) int64|int32 — reverse moves are rewritten in place (:221-222)
 *   reward          out (N) fp32; done, self_collision, edge_collision out (N) bytes
 *   obs             out, wurm_single_obs_elems() floats per env (ignored for WURM_OBS_NONE); the observation
 *                   of the post-step, pre-reset state (:304)
 *   inject_food     nullable (N) int32: flat cell (y*S+x) the respawned food goes to when env i eats, -1 = none */
int wurm_single_step(float *envs, void *actions, int actions_dtype, float *reward, uint8_t *done,
                     uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                     int64_t num_envs, int size, uint64_t seed, uint64_t call, int64_t env_offset,
                     const int32_t *inject_food, void *stream);

/* SingleSnake.reset (single_snake.py:322-342) + _create_envs (:344-387): envs with done[i] != 0 are rebuilt
 * (3-segment snake at a random seed cell/direction, one food), then every env is observed.
 *   inject_reset    nullable (N,4) int32: seed_y, seed_x, direction, food_cell
 * Returns WURM_ERR_UNSUPPORTED for size <= 8 (:346-347). */
int wurm_single_reset(float *envs, const uint8_t *done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                      int size, uint64_t seed, uint64_t call, int64_t env_offset, const int32_t *inject_reset,
                      void *stream);

/* SingleSnake._observe (single_snake.py:130-195) */
int wurm_single_observe(const float *envs, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                        void *stream);

/* T fused iterations of the caller loop of tests/test_single_snake_env.py:24-31 / experiments/main.py:212-227:
 *   for t: outputs[t] = step(actions[t]) with call = call0 + 2t;  reset(done[t]) with call = call0 + 2t + 1
 * in ONE launch with the env resident on chip.  actions (T,N) in/out; reward/done/... (T,N); obs (T,N,elems);
 * inject_food (T,N), inject_reset (T,N,4) nullable.  Bit-identical to T calls of the two entry points above. */
int wurm_single_rollout(float *envs, void *actions, int actions_dtype, float *reward, uint8_t *done,
                        uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                        int64_t num_envs, int size, int64_t num_steps, uint64_t seed, uint64_t call0,
                        int64_t env_offset, const int32_t *inject_food, const int32_t *inject_reset, void *stream);

/* wurm_single_rollout (RNG mode) for a caller that keeps the mirror of wurm_single_call.resident (wurm_single_resident_bytes;
 * *resident_valid / resident_lazy as the fields of wurm_single_call).  Grids of 12 x 12 and larger: an env the mirror's record
 * describes is read from its clock grid (2 bytes per cell instead of 12) and written back there — the planes only while the
 * mirror is not lazy — and *resident_valid = 1 afterwards (round 6: a 16-step launch of BASELINE configs[4] no longer reads
 * 127 MB of planes).  9 x 9 and every launch the clock-grid rollout does not serve: wurm_single_rollout on the planes after
 * writing a lazy valid mirror out; *resident_valid = 0.  resident == NULL: wurm_single_rollout. */
int wurm_single_rollout_resident(float *envs, void *actions, int actions_dtype, float *reward, uint8_t *done,
                                 uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                                 int64_t num_envs, int size, int64_t num_steps, uint64_t seed, uint64_t call0,
                                 int64_t env_offset, void *resident, int *resident_valid, int resident_lazy, void *stream);

/* Arguments of wurm_single_step_reset / wurm_grid_step_reset (a HOST struct of device pointers and sizes; the fields
 * shared with wurm_single_step / wurm_single_reset mean the same). */
typedef struct wurm_single_call {
    float *envs;                     /* in/out (N,3,S,S) [SimpleGridworld: (N,2,S,S)]                            */
    void *actions;                   /* in/out (N) int64|int32 (SimpleGridworld: read only)                      */
    float *reward;                   /* out (N)                                                                  */
    uint8_t *done;                   /* out (N)                                                                  */
    uint8_t *self_collision;         /* out (N), SingleSnake only                                                */
    uint8_t *edge_collision;         /* out (N)                                                                  */
    float *obs;                      /* out: observation of the post-step, PRE-reset state (single_snake.py:304) */
    float *obs_after;                /* nullable out: what reset(done) returns (:342) — the observation of every
                                        env once the envs that just finished are rebuilt with call + 1          */
    uint8_t *done_copy;              /* nullable out (N): second copy of `done`                                  */
    const uint8_t *pre_done;         /* nullable in (N): envs flagged here are rebuilt BEFORE the step, exactly as
                                        wurm_single_reset(done = pre_done, call = pre_call) would               */
    const int32_t *inject_food;      /* nullable, as wurm_single_step                                            */
    const int32_t *inject_reset;     /* nullable (N,4) [grid: (N)]: outcomes of the reset that follows the step  */
    const int32_t *inject_pre_reset; /* nullable (N,4) [grid: (N)]: outcomes of the reset in front of the step   */
    int64_t num_envs;
    int64_t env_offset;
    uint64_t seed;
    uint64_t call;                   /* counter of the step; the reset that follows it uses call + 1             */
    uint64_t pre_call;               /* counter of the reset in front of the step                                */
    int actions_dtype, obs_mode, obs_n, size;
    int post_reset;                  /* != 0: envs that finished are rebuilt (call + 1) and STORED after `obs`   */
    int start_y, start_x;            /* SimpleGridworld start location (simple_gridworld.py:254-262)             */
    void *resident;                  /* nullable in/out: wurm_single_resident_bytes() (SimpleGridworld:
                                        wurm_grid_resident_bytes()) bytes the caller owns — a compact mirror of `envs` that
                                        the step reads INSTEAD of envs and keeps current (envs itself is still written
                                        every call)                                                                */
    int resident_valid;              /* != 0: nothing but calls that were given `resident` has written envs since
                                        the mirror was last maintained; 0: it is rebuilt from envs first.
                                        SimpleGridworld also knows 2 = REFUSED: the launch that built the mirror found envs
                                        outside the lane kernel's domain; the planes stay the state and the mirror unused
                                        until the caller clears the field again (wurm_grid_resident_bytes)          */
    int resident_lazy;               /* != 0: the step does not write envs at all; envs is brought up to date by
                                        wurm_single_resident_flush (call it before anything else reads or writes
                                        envs, and before clearing resident_valid)                                 */
    uint32_t *check_mask;            /* nullable out (N), SingleSnake: wurm_single_check's mask of the POST-STEP state of
                                        every env, computed inside the step launch where the launch holds all there is to
                                        check (the resident step: an env in its domain is a well-formed snake — what is
                                        left to say is WURM_CHK_MIN_LENGTH / WURM_CHK_ONE_FOOD); WURM_CHK_NOT_COMPUTED for
                                        an env that finished in this step or that the launch cannot vouch for, and for
                                        every env when another kernel served the call.  experiments/main.py:214-215 checks
                                        `env.envs[~done]` every step: with this the check is one any() over N words       */
} wurm_single_call;

/* One launch for one iteration of the caller loop of tests/test_single_snake_env.py:24-31 /
 * experiments/main.py:212-227,
 *     obs, reward, done, info = env.step(actions);  env.reset(done)
 * in either grouping: post_reset != 0 = [step, observe, reset] (what wurm_single_rollout does per iteration); pre_done
 * given = [the reset the caller postponed, step, observe] — the host class defers reset(done) into the next step's
 * launch and flushes it with wurm_single_reset if the state is looked at in between, so `envs` is always what the
 * reference would show.  Bit-identical to the wurm_single_step / wurm_single_reset pair with the same counters. */
int wurm_single_step_reset(const wurm_single_call *c, void *stream);

/* Size in bytes of the mirror of wurm_single_call.resident for this batch, 0 if the shape is not served by it (then pass
 * resident = NULL).  Served: size 9 with observation none or partial_2 from 4096 envs on (32 bytes per env,
 * wurm_amd/csrc/lane_resident.hpp), sizes 10 and 11 with observation default, one_channel, partial_2, partial_3, positions or
 * none from 4096 envs on (48 bytes per env, lane_wide_resident.hpp; maintained by calls with resident_lazy != 0 only: a call with
 * resident_lazy == 0 steps envs without the mirror and leaves it stale — wurm_single_step_slot clears resident_valid, a caller
 * of wurm_single_step_reset does so itself), and 12 <= size <= 64 with any observation from 2^20 cells in the batch on (the 16-bit
 * clock grid of the LDS step + 48 bytes per env, grid_rollout.hip); WURM_RESIDENT_MIN_ENVS replaces the thresholds by
 * a number of envs.  With the mirror the per-call step of a large batch does not read the (N,3,S,S) state at all.  Protocol: a call of
 * wurm_single_step_reset with `resident` given leaves the mirror current unless it had inject_* / post_reset set;
 * wurm_single_step_slot maintains c->resident_valid itself after each call, the caller only CLEARS it whenever anything
 * else writes `envs` (another entry point, the caller's own code).  A call that cannot use the mirror (inject_* /
 * post_reset) flushes a lazy mirror into envs first by itself. */
int64_t wurm_single_resident_bytes(int64_t num_envs, int size, int obs_mode, int obs_n);
/* the same without the batch-size threshold: the size of the mirror whenever the SHAPE is served (a caller that asked for
 * the mirror explicitly, `resident_mirror=True` of the Python classes), 0 if it is not */
int64_t wurm_single_resident_size(int64_t num_envs, int size, int obs_mode, int obs_n);

/* resident_lazy: writes envs from the mirror (every env the mirror describes, whole; the others — states outside the
 * lane kernels' domain, which the step keeps in envs itself — are left as they are).  A no-op returning WURM_OK unless
 * c->resident, c->resident_lazy and c->resident_valid are all set.  Leaves the mirror valid. */
int wurm_single_resident_flush(const wurm_single_call *c, void *stream);

/* The same for SimpleGridworld (simple_gridworld.py:135-202,225-268). */
int wurm_grid_step_reset(const wurm_single_call *c, void *stream);

/* SimpleGridworld's mirror (round 6; wurm_amd/csrc/gridworld_lane.hip): 16 bytes of header + one 32-bit record per env (agent
 * cell | food cell << 16) — an env in the reference's own invariant IS two cell indices, so with the mirror the per-call
 * step of a large batch neither scans the (N,2,S,S) planes nor (while the mirror is lazy) writes them, and it is ONE launch:
 * the lane kernel's domain (at most one agent, at most one food, values 0 / 1) is closed under this library's own launches,
 * so once a launch has built the mirror without meeting an env outside it, none can appear until something else writes the
 * state — which is when the caller clears resident_valid.  The launch that BUILDS the mirror (resident_valid == 0) reads
 * that verdict back synchronously (4 bytes, one stream synchronisation): resident_valid becomes 1, or 2 = refused (then the
 * planes are the state, every call takes the two-launch path, and the mirror is tried again when the caller clears the
 * field).  Served: 5 <= size <= 64, observation default / raw / positions / none, from WURM_LANE_STEP_MIN_ENVS envs on
 * (WURM_RESIDENT_MIN_ENVS overrides); wurm_grid_resident_size: without the batch-size threshold.  Protocol otherwise as
 * wurm_single_resident_bytes: wurm_grid_step_slot maintains c->resident_valid (wurm_grid_step_reset, whose block is const,
 * returns WURM_MIRROR_REFUSED instead of WURM_OK from the call that was to build the mirror: the caller sets the field to 2),
 * the caller clears it whenever anything else writes envs, and wurm_grid_resident_flush writes envs from a lazy valid mirror
 * (a no-op otherwise).  0 also for an image observation whose run of four envs exceeds the lane kernel's 16 KB byte slab. */
int64_t wurm_grid_resident_bytes(int64_t num_envs, int size, int obs_mode);
int64_t wurm_grid_resident_size(int64_t num_envs, int size, int obs_mode);
int wurm_grid_resident_flush(const wurm_single_call *c, void *stream);

/* Caller-allocated output slabs holding the fresh output tensors of the next `steps` step() calls (the host classes
 * carve the tensors they return out of them; a slab is never written twice).  Layouts: obs, obs_after
 * (steps, N, elems) fp32; reward (steps, N) fp32; flags (3, steps, N) bytes = done, self_collision, edge_collision. */
typedef struct wurm_single_slabs {
    float *obs;
    float *obs_after; /* nullable */
    float *reward;
    uint8_t *flags;
    int64_t steps;
} wurm_single_slabs;

/* wurm_single_step_reset with the per-call bookkeeping of the host class done here instead of in Python (at 512 envs
 * the loop of experiments/main.py:212-227 is bound by host time per call): fills c->actions / actions_dtype / call and
 * the output pointers of slot `slot` of the slabs (obs_after only if want_obs_after), sets c->pre_done = c->done_copy
 * and c->pre_call when `apply_pending` (the reset the caller postponed), else clears c->pre_done, and launches.
 * Everything else in *c (envs, sizes, seed, modes, done_copy, injections) is left as the caller set it. */
int wurm_single_step_slot(wurm_single_call *c, const wurm_single_slabs *slabs, int64_t slot, void *actions,
                          int actions_dtype, uint64_t call, int apply_pending, uint64_t pre_call, int want_obs_after,
                          void *stream);

/* The same for SimpleGridworld (the self_collision plane of `flags` is not written). */
int wurm_grid_step_slot(wurm_single_call *c, const wurm_single_slabs *slabs, int64_t slot, void *actions,
                        int actions_dtype, uint64_t call, int apply_pending, uint64_t pre_call, int want_obs_after,
                        void *stream);

/* The acting half of the single-agent loop, experiments/main.py:207-212,227, T iterations in ONE launch:
 *   probs, value = model(state)            FeedforwardAgent, wurm/agents/feedforward.py:8-28: E -> 64 -> 64 -> {4, 1}
 *   action = Categorical(probs).sample()   main.py:208-210
 *   state, reward, done, info = env.step(action);  env.reset(done)     (`state`: the PRE-reset observation, :212)
 * obs0 (N,E): the observation the policy acts on at step 0 (the caller's `state`), E = 3 (2 obs_n + 1)^2.
 * params: W1 (64,E) b1 (64) W2 (64,64) b2 (64) Wp (4,64) bp (4) Wv (64) bv (1), fp32, torch Linear layout,
 * concatenated.  Outputs (T,N,..): actions int64 (sanitised by step as in the reference), probs (4), values, reward,
 * done, collision flags, obs (E) = the observation step t returned = the policy input of step t+1.
 * status (N) uint8: 0 = rolled out; 1 = the env's state is outside the kernel's domain (not a well-formed snake):
 * it is left untouched and its outputs are not written.  Step t uses call0 + 2t (env step and the sampling draw),
 * its reset call0 + 2t + 1.  Arithmetic order of the policy: see wurm_amd/csrc/policy_rollout.hpp.
 * Supported: 9 <= size <= 11, 0 <= obs_n <= 3. */
int wurm_single_policy_rollout(float *envs, const float *obs0, const float *params, int64_t *actions, float *probs,
                               float *values, float *reward, uint8_t *done, uint8_t *self_collision,
                               uint8_t *edge_collision, float *obs, uint8_t *status, int obs_n, int64_t num_envs,
                               int size, int64_t num_steps, uint64_t seed, uint64_t call0, int64_t env_offset,
                               void *stream);

/* wurm.utils.env_consistency / snake_consistency (wurm/utils.py:113-178) as a per-env error bitmask
 * (bit i = i-th check failed, WURM_CHK_*; 0 = consistent).  err out (N) uint32. */
#define WURM_CHK_FOOD_VALUE 1u
#define WURM_CHK_ONE_HEAD 2u
#define WURM_CHK_HAS_SNAKE 4u
#define WURM_CHK_HEAD_AT_END 8u
#define WURM_CHK_BODY_RANGE 16u
#define WURM_CHK_MIN_LENGTH 32u
#define WURM_CHK_HEAD_ON_FOOD 64u
#define WURM_CHK_ONE_FOOD 128u
int wurm_single_check(const float *envs, uint32_t *err, int64_t num_envs, int size, void *stream);

/* ------------------------------------------------------------------------------------------- SimpleGridworld */

/* SimpleGridworld.step (simple_gridworld.py:135-202) + _observe (:111-133); actions are read only. */
int wurm_grid_step(float *envs, const void *actions, int actions_dtype, float *reward, uint8_t *done,
                   uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                   uint64_t seed, uint64_t call, int64_t env_offset, const int32_t *inject_food, void *stream);

/* SimpleGridworld.reset (simple_gridworld.py:225-268): agent at (start_y,start_x), one food.
 *   inject_reset nullable (N) int32 food cell.  WURM_ERR_UNSUPPORTED for size <= 4 or no start location (:249-260). */
int wurm_grid_reset(float *envs, const uint8_t *done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                    int size, int start_y, int start_x, uint64_t seed, uint64_t call, int64_t env_offset,
                    const int32_t *inject_reset, void *stream);

int wurm_grid_observe(const float *envs, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                      void *stream);

int wurm_grid_rollout(float *envs, const void *actions, int actions_dtype, float *reward, uint8_t *done,
                      uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                      int64_t num_steps, int start_y, int start_x, uint64_t seed, uint64_t call0,
                      int64_t env_offset, const int32_t *inject_food, const int32_t *inject_reset, void *stream);

/* wurm_grid_rollout (RNG mode) for a caller that keeps SimpleGridworld's mirror (wurm_grid_resident_bytes; *resident_valid and
 * resident_lazy as the fields of wurm_single_call): where the one-env-per-lane kernel serves the launch, the state comes from
 * the records when *resident_valid == 1 — no scan of the planes, no flag pass behind the launch, and (lazy) no write to the
 * planes — and the records describe the final state afterwards.  A launch that BUILDS the mirror reads its verdict back
 * (one stream synchronisation): *resident_valid = 1, or 2 = refused.  Any other launch runs wurm_grid_rollout on the planes
 * after writing a lazy valid mirror out, and leaves *resident_valid = 0 (2 stays 2).  resident == NULL: wurm_grid_rollout. */
int wurm_grid_rollout_resident(float *envs, const void *actions, int actions_dtype, float *reward, uint8_t *done,
                               uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                               int64_t num_steps, int start_y, int start_x, uint64_t seed, uint64_t call0, int64_t env_offset,
                               void *resident, int *resident_valid, int resident_lazy, void *stream);

/* ------------------------------------------------------------------------------------------- MultiSnake */

/* Dynamics parameters of MultiSnake (attributes the reference lets callers change after construction,
 * multi_snake.py:121-129; tests/test_multi_snake_env.py:180,288,401-403 set them).  A HOST struct. */
typedef struct wurm_multi_config {
    int boost;             /* self.boost                                           multi_snake.py:123 */
    int food_on_death;     /* self.food_on_death_prob > 0                          :565,662           */
    float death_threshold; /* (float)(1 - food_on_death_prob): food where u > threshold   :424        */
    float boost_cost_prob; /* boost cost where u < boost_cost_prob                 :579               */
    int food_mode;         /* 0 = 'only_one', 1 = 'random_rate'                    :369,380           */
    float food_rate;       /* rate food where u < food_rate                        :403               */
    int max_food;          /* num_snakes * 8                                       :127               */
    float reward_on_death; /*                                                      :684               */
    int respawn_any;       /* respawn_mode == 'any'                                :805               */
    int colour_random;     /* colour_mode == 'random'                              :800               */
} wurm_multi_config;

/* Recorded random outcomes for wurm_multi_step (all DEVICE pointers; pass the struct pointer as NULL for RNG mode). */
typedef struct wurm_multi_inject {
    const uint8_t *death_a;   /* (N,S,S) outcome of `rand_like > 1-p`, boost phase    :424 via :574 */
    const uint8_t *cost;      /* (N*K)   outcome of `rand < boost_cost_prob`          :579          */
    const uint8_t *death_b;   /* (N,S,S) outcome of `rand_like > 1-p`, regular phase  :424 via :671 */
    const uint8_t *rate;      /* (N,S,S) outcome of `rand < food_rate`                :401-403      */
    const int32_t *food_cell; /* (N)     'only_one' respawn cell, -1 = none           :374-379      */
} wurm_multi_inject;

/* Recorded random outcomes for wurm_multi_reset (DEVICE pointers; NULL struct pointer = RNG mode). */
typedef struct wurm_multi_reset_inject {
    const int32_t *create;      /* (N,K,2) seed cell, direction of every snake of a rebuilt env    :996-1019 */
    const int32_t *create_food; /* (N)     food cell of a rebuilt env                               :1016     */
    const int16_t *colours;     /* (N*K,3) colour given to snakes still dead after the reset        :800-803  */
    const int32_t *respawn;     /* (N,2)   respawn_mode 'any': seed cell (-1 = no room), direction  :805-829  */
} wurm_multi_reset_inject;

/* floats per (agent, env) of an observation: 'full' (WURM_OBS_DEFAULT) 3*S*S, 'partial_n' 3*(2n+1)^2 */
int64_t wurm_multi_obs_elems(int obs_mode, int obs_n, int size);

/* MultiSnake.step (multi_snake.py:462-731) + _observe (:283-334) in one launch.
 *   foods (N,1,S,S), heads/bodies (N*K,1,S,S) fp32 in/out; dones (N*K) bytes in/out; orientations (N*K) int64
 *   in/out; actions (K,N) int64: actions[i*N + e] = agent_i's action (0..7) in env e (the reference stacks its
 *   dict the same way, :492); colours (N*K,3) int16 (agent_colours, read by 'partial_n').
 *   Outputs, (N*K) in the reference's agent order env*K + i: boost_this_step, rewards, snake_collision,
 *   edge_collision, food_consumed (info 'food_i'), sizes (info 'size_i'); all_done (N) = dones['__all__'].
 *   obs (K,N,elems): obs[i] is agent_i's tensor.
 *   agent_major_f32 / agent_major_u8: nullable.  The same per-agent outputs transposed to agent-major (K,N) rows, so
 *   that the per-agent dicts of :701-729 are plain row views: f32 (3,K,N) = rewards, food_consumed, sizes;
 *   u8 (4,K,N) = dones, boost_this_step, snake_collision, edge_collision. */
int wurm_multi_step(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                    const int64_t *actions, uint8_t *boost_this_step, float *rewards, uint8_t *snake_collision,
                    uint8_t *edge_collision, float *food_consumed, float *sizes, uint8_t *all_done,
                    const int16_t *colours, float *obs, int obs_mode, int obs_n, int64_t num_envs, int num_snakes,
                    int size, const wurm_multi_config *cfg, uint64_t seed, uint64_t call, int64_t env_offset,
                    const wurm_multi_inject *inject, float *agent_major_f32, uint8_t *agent_major_u8, void *stream);

/* Arguments of wurm_multi_step_reset: the pointers and sizes of wurm_multi_step (same meaning), plus the postponed
 * reset.  A HOST struct of device pointers. */
typedef struct wurm_multi_call {
    float *foods, *heads, *bodies;
    uint8_t *dones;
    int64_t *orientations;
    int16_t *colours;                 /* in/out when pre_done is given (re-rolled colours), else read ('partial_n')   */
    const int64_t *actions;           /* (K,N)                                                                         */
    uint8_t *boost_this_step;
    float *rewards;
    uint8_t *snake_collision, *edge_collision;
    float *food_consumed, *sizes;
    uint8_t *all_done;                /* out (N) = dones['__all__']                                                    */
    uint8_t *all_done_copy;           /* nullable out (N): second copy of all_done                                     */
    float *obs;
    float *obs_after;                 /* nullable out, shaped like obs: what wurm_multi_reset(done_env = all_done,
                                         call = call + 1) would return — the observation of every agent after the
                                         envs that just finished are rebuilt, dead snakes' colours re-rolled and
                                         (respawn_any) one dead snake respawned.  Only written: that reset is NOT
                                         applied to the state tensors (pass all_done_copy as the next pre_done)         */
    float *agent_major_f32;           /* nullable, as wurm_multi_step                                                  */
    uint8_t *agent_major_u8;
    const uint8_t *pre_done;          /* nullable in (N): wurm_multi_reset(done_env = pre_done, call = pre_call, no
                                         observation) is applied to every env BEFORE the step                         */
    const wurm_multi_inject *inject;            /* nullable: outcomes of the step                                      */
    const wurm_multi_reset_inject *pre_inject;  /* nullable: outcomes of the reset in front of it                      */
    int64_t num_envs, env_offset;
    uint64_t seed, call, pre_call;
    int num_snakes, size, obs_mode, obs_n;
    wurm_multi_config cfg;
    void *resident;                   /* nullable in/out: wurm_multi_resident_bytes() bytes the caller owns — a compact
                                         mirror of foods / heads / bodies that the step reads INSTEAD of them and keeps
                                         current (dones, orientations, colours are always read and written in place)      */
    int resident_valid;               /* != 0: nothing but calls that were given `resident` has written foods / heads /
                                         bodies since the mirror was last maintained; 0: the step reads them            */
    int resident_lazy;                /* != 0: the step does not write foods / heads / bodies; wurm_multi_resident_flush
                                         brings them up to date (before anything else reads or writes them, and before
                                         resident_valid is cleared)                                                     */
    uint32_t *check_mask;             /* nullable out (N): wurm_multi_check's mask of the post-step state, computed from
                                         the on-chip image inside the step launch — for an env whose image came from the
                                         mirror or from a rebuild; 0xffffffff = not computed (the env was read from the
                                         fp32 tensors, which can hold more than the image: run wurm_multi_check)         */
    uint32_t *check_mask_after;       /* nullable out (N), with obs_after: the same for the state obs_after observes   */
} wurm_multi_call;

/* One launch for one iteration of the caller loop of experiments/speeds.py:30-37 / tests/test_multi_snake_env.py:78-89,
 *     obs, rewards, dones, info = env.step(actions);  env.reset(dones['__all__'])
 * as [the reset the caller postponed, step, observe]: the host class defers reset(...) into the next step's launch and
 * flushes it with wurm_multi_reset if the state is looked at in between.  Bit-identical to the wurm_multi_reset (without
 * observation) / wurm_multi_step pair with the same counters.  pre_done = NULL: plain wurm_multi_step. */
int wurm_multi_step_reset(const wurm_multi_call *c, void *stream);

/* wurm_multi_step_reset for a host loop that keeps ONE call block alive (state pointers, sizes, seed, configuration,
 * all_done_copy) and hands over only what changes from step to step; the output pointers of the block are filled from
 * two packed buffers (so that the Python class allocates twice per step, not six times, and unbinds them into the
 * per-agent dicts of multi_snake.py:701-729); obs_after (nullable) as in wurm_multi_call:
 *   out_f32 (6*K*N floats): [0,KN) rewards, [KN,2KN) food_consumed, [2KN,3KN) sizes, env-major [env*K + i];
 *                           then the same three agent-major, (3,K,N), [i*N + env]
 *   out_u8 (7*K*N + N bytes): boost_this_step, snake_collision, edge_collision env-major; then agent-major (4,K,N)
 *                           dones, boost, snake_collision, edge_collision; then all_done (N)
 *   apply_pending != 0: the postponed reset with pre_done = c->all_done_copy (the previous step's all_done) and
 *                           counter pre_call goes in front of the step.
 * c->inject / c->pre_inject are used as they stand.  Bit-identical to wurm_multi_step_reset on the same pointers. */
int wurm_multi_step_packed(wurm_multi_call *c, float *out_f32, uint8_t *out_u8, float *obs, float *obs_after,
                           const int64_t *actions, uint64_t call, int apply_pending, uint64_t pre_call, void *stream);

/* Caller-allocated output slabs holding the fresh output tensors of the next `steps` step() calls of MultiSnake (as
 * wurm_single_slabs: the host class carves the tensors of its per-agent dicts, multi_snake.py:701-729, out of them ONCE per
 * slab; a slab is never written twice).  Per step the layouts of wurm_multi_step_packed:
 * out_f32 (steps, 6 K N) floats; out_u8 (steps, 7 K N + N) bytes; obs, obs_after (steps, K, N, elems) floats. */
typedef struct wurm_multi_slabs {
    float *out_f32;
    uint8_t *out_u8;
    float *obs;
    float *obs_after; /* nullable */
    int64_t steps;
    int64_t obs_elems; /* floats per (agent, env) of an observation: wurm_multi_obs_elems(c->obs_mode, c->obs_n, c->size) */
} wurm_multi_slabs;

/* wurm_multi_step_packed on slot `slot` of the slabs (obs_after only if want_obs_after): one C call per
 * `env.step(actions)` with nothing to allocate or to compute on the host (wurm_amd/csrc/fastcall.c: Stepper.step_multi). */
int wurm_multi_step_slot(wurm_multi_call *c, const wurm_multi_slabs *slabs, int64_t slot, const int64_t *actions,
                         uint64_t call, int apply_pending, uint64_t pre_call, int want_obs_after, void *stream);

/* The mirror of wurm_multi_call.resident: per env the 16-bit clock grids of the K bodies, the food grid as bytes and three
 * ints per snake — (2 K + 1) S^2 bytes and change instead of (1 + 2 K) S^2 fp32 read every call
 * (wurm_amd/csrc/multi_snake.hip).  Returns its size in bytes, 0 if it is not offered for this batch (fewer than 2^20
 * cells num_envs * num_snakes * size^2; WURM_RESIDENT_MIN_ENVS replaces that by a number of envs).  Protocol as for
 * wurm_single_call.resident: wurm_multi_step_packed sets c->resident_valid after a launch that maintained the mirror
 * (not with inject / pre_inject: such a call writes a lazy mirror out first and steps the fp32 state), the caller clears it
 * whenever anything else writes the state — after wurm_multi_resident_flush if the mirror is lazy. */
int64_t wurm_multi_resident_bytes(int64_t num_envs, int num_snakes, int size);
int64_t wurm_multi_resident_size(int64_t num_envs, int num_snakes, int size); /* without the batch-size threshold */
int wurm_multi_resident_flush(const wurm_multi_call *c, void *stream);

/* MultiSnake.reset (multi_snake.py:771-836): envs flagged in done_env (N bytes) are rebuilt (_create_envs
 * :996-1019: K snakes placed one after another on free cells away from everything, one food); colours of
 * snakes that are still dead are re-rolled (colour_random); respawn_any: the first dead snake of every env
 * respawns if there is room; then every agent is observed (obs may be NULL / WURM_OBS_NONE).
 *   status: nullable, 1 int32, incremented for every env in which a rebuilt snake found no room (the reference
 *   raises RuntimeError at :946-947). */
int wurm_multi_reset(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                     int16_t *colours, const uint8_t *done_env, int32_t *status, const uint8_t *boost_this_step,
                     float *obs, int obs_mode, int obs_n, int64_t num_envs, int num_snakes, int size,
                     const wurm_multi_config *cfg, uint64_t seed, uint64_t call, int64_t env_offset,
                     const wurm_multi_reset_inject *inject, void *stream);

/* T fused iterations of the caller loop of experiments/speeds.py:30-37 / tests/test_multi_snake_env.py:78-89:
 *   for t: outputs[t], obs[t] = step(actions[t]) with call = call0 + 2t;  reset(dones['__all__']) with call0 + 2t + 1
 * in ONE launch with the env resident on chip (LDS).  Bit-identical to T calls of wurm_multi_step / wurm_multi_reset
 * (the latter without observation).
 *   actions (T,K,N) int64;  out_f32 (T,3,K,N) = rewards, food_consumed, sizes;  out_u8 (T,4,K,N) = dones,
 *   boost_this_step, snake_collision, edge_collision (agent-major rows, as the agent_major_* outputs of
 *   wurm_multi_step);  all_done (T,N);  obs (T,K,N,elems).  State tensors, dones, orientations, colours and
 *   boost_this_step (nullable) are updated in place to the state after the last reset.
 *   inject / reset_inject: both NULL (RNG mode) or both given, each array with a leading T dimension. */
int wurm_multi_rollout(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                       int16_t *colours, uint8_t *boost_this_step, const int64_t *actions, float *out_f32,
                       uint8_t *out_u8, uint8_t *all_done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                       int num_snakes, int size, int64_t num_steps, const wurm_multi_config *cfg, uint64_t seed,
                       uint64_t call0, int64_t env_offset, const wurm_multi_inject *inject,
                       const wurm_multi_reset_inject *reset_inject, void *stream);

/* wurm_multi_rollout (RNG mode) for a caller that keeps the compact mirror of wurm_multi_call (`resident`,
 * wurm_multi_resident_bytes() bytes; same meaning of *resident_valid / resident_lazy as there): where the launch is served by
 * the kernel that can keep it ('full' observations of at most 10 snakes, more than one step, a large batch) the state is
 * read from the mirror when *resident_valid != 0 — 2 K + 1 bytes per cell instead of (1 + 2 K) fp32 planes — and the mirror
 * describes the final state afterwards (*resident_valid = 1); foods / heads / bodies are then written only if
 * resident_lazy == 0.  Any other shape runs wurm_multi_rollout on the fp32 planes: a lazy valid mirror is written out to
 * them first (wurm_multi_resident_flush) and *resident_valid = 0 afterwards.  resident == NULL: wurm_multi_rollout.
 * Replaces nothing in the reference (MultiSnake has no fused loop); it is to `env.rollout` what wurm_multi_step_packed's
 * mirror is to `env.step` (multi_snake.py:462-731, 771-836 over T iterations). */
int wurm_multi_rollout_resident(float *foods, float *heads, float *bodies, uint8_t *dones, int64_t *orientations,
                                int16_t *colours, uint8_t *boost_this_step, const int64_t *actions, float *out_f32,
                                uint8_t *out_u8, uint8_t *all_done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                                int num_snakes, int size, int64_t num_steps, const wurm_multi_config *cfg, uint64_t seed,
                                uint64_t call0, int64_t env_offset, void *resident, int *resident_valid, int resident_lazy,
                                void *stream);

/* MultiSnake._observe (multi_snake.py:283-334) */
int wurm_multi_observe(const float *foods, const float *heads, const float *bodies, const uint8_t *dones,
                       const uint8_t *boost_this_step, const int16_t *colours, float *obs, int obs_mode, int obs_n,
                       int64_t num_envs, int num_snakes, int size, void *stream);

/* MultiSnake.check_consistency (multi_snake.py:733-769) as a per-env bitmask: bits 0-6 = WURM_CHK_* of any living
 * snake, WURM_MCHK_OVERLAP, WURM_MCHK_DEAD_NONZERO. */
#define WURM_CHK_NOT_COMPUTED 0xFFFFFFFFu /* wurm_single_call.check_mask / wurm_multi_call.check_mask: run the checker */
#define WURM_MCHK_OVERLAP 0x100u
#define WURM_MCHK_DEAD_NONZERO 0x200u
int wurm_multi_check(const float *foods, const float *heads, const float *bodies, const uint8_t *dones, uint32_t *err,
                     int64_t num_envs, int num_snakes, int size, void *stream);

/* MultiSnake.get_n_colours (multi_snake.py:163-169) at construction: random colour per agent (fixed = 0) or one
 * colour per snake index shared by all envs (fixed = 1; :146-148).  colours out (N*K,3) int16. */
int wurm_multi_colours(int16_t *colours, int64_t num_envs, int num_snakes, int fixed, uint64_t seed, uint64_t call,
                       int64_t env_offset, void *stream);

/* wurm.utils.determine_orientations (wurm/utils.py:36-65) over a (n,3,S,S) batch; out (n) int64. */
int wurm_orientations(const float *envs, int64_t *out, int64_t n, int size, void *stream);

/* ------------------------------------------------------------------------------------------- learner-side glue
 * (the consumers of the env's outputs in experiments/main.py) */

/* Return computation of wurm.rl.A2C.loss (wurm/rl/a2c.py:49-66): reverse scan over time per env.
 *   rewards, values, returns (T,N) fp32 row-major; dones (T,N) bytes; bootstrap (N) fp32.
 *   use_gae = 0: R = bootstrap * !done[T-1]; R_t = r_t + gamma * R_{t+1} * !done_t            (:60-64; values unused)
 *   use_gae = 1: delta_t = r_t + gamma * v_{t+1} * !done_t - v_t; gae_t = delta_t + gamma_lambda * !done_t * gae_{t+1};
 *                R_t = gae_t + v_t, with gamma_lambda = (float)(gamma * gae_lambda)            (:50-59)
 * fp32 in the reference's operation order: bit-identical to its torch-CPU path. */
int wurm_a2c_returns(const float *bootstrap, const float *rewards, const float *values, const uint8_t *dones,
                     float gamma, int use_gae, float gamma_lambda, float *returns, int64_t num_steps, int64_t num_envs,
                     void *stream);

/* Gradient of the scan above: given grad_returns (T,N), writes grad_values (T,N; nullable) and grad_bootstrap (N;
 * nullable) — what torch autograd computes through the reference's op-by-op construction of `returns`. */
int wurm_a2c_returns_backward(const float *grad_returns, const uint8_t *dones, float gamma, int use_gae,
                              float gamma_lambda, float *grad_values, float *grad_bootstrap, int64_t num_steps,
                              int64_t num_envs, void *stream);

/* The per-step logging reductions of experiments/main.py:252-274 in one launch: adds to accum (5 doubles on the
 * device, zeroed by the caller) the batch sums of done, reward, edge_collision, self_collision and snake length
 * (max of the body channel of envs (N,3,S,S)). */
int wurm_single_stats(const float *envs, const float *reward, const uint8_t *done, const uint8_t *self_collision,
                      const uint8_t *edge_collision, double *accum, int64_t num_envs, int size, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WURM_HIP_H */
